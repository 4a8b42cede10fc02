"""Where the split kernel stops winning: iteration time with split_kernel = 1 and -1 over a grid of shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
shapes = [(4096, 16384, 128), (4096, 8192, 128), (4096, 4096, 128), (4096, 2048, 128), (4096, 1024, 128), (2048, 16384, 128), (1024, 16384, 128),
          (4096, 16384, 64), (4096, 8192, 64), (4096, 4096, 64), (4096, 2048, 64), (2048, 8192, 64), (1024, 16384, 64), (1024, 65536, 64), (512, 65536, 32), (8192, 4096, 32),
          (1024, 4096, 256), (4096, 350, 256), (4096, 1024, 256), (2048, 4096, 256), (4096, 4096, 256), (1024, 1024, 200)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    t = {}
    for sk in (1, -1):
        s = ng.Solver(M, N, K, split_kernel=sk)
        s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)), np.asfortranarray(rng.random((M, N), dtype=np.float32)))
        s.iterate(41); s.sync()
        t0 = time.perf_counter(); s.iterate(64); s.sync(); t[sk] = (time.perf_counter() - t0) / 64
        s.close()
    f = 8.0 * M * N * K
    print(f"({M},{N},{K}) M*N=2^{np.log2(M * N):.1f}: split {t[1] * 1e6:8.1f} us = {f / t[1] / 1e12:6.1f} TF | 64-column {t[-1] * 1e6:8.1f} us = {f / t[-1] / 1e12:6.1f} TF | ratio {t[-1] / t[1]:.2f}", flush=True)
