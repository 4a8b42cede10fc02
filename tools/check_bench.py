"""The KL / rel-L1 check at cfg3: the dominant kernel (product 1 + one log2 per element) and the whole check() call
(that kernel, the fp64 factor sums, the composition, the copy of three doubles and the synchronisation)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 65536, 256)
rng = np.random.default_rng(0)
s = ng.Solver(M, N, K)
s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)), np.asfortranarray(rng.random((M, N), dtype=np.float32)))
s.iterate(1); kl = s.check()
t0 = time.perf_counter()
for _ in range(20): s.check()
dt = (time.perf_counter() - t0) / 20
print(f"({M},{N},{K}) check: {kl}  kernel {s.time_piece(6, 10):.3f} ms, whole check() call {dt * 1e3:.3f} ms")
