"""Soak: 10 000 iterations at cfg3 (or M N K from the command line) with a KL check every 1000 (monotone, finite); ~40 s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 65536, 256)
rng = np.random.default_rng(0)
s = ng.Solver(M, N, K)
s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)), np.asfortranarray(rng.random((M, N), dtype=np.float32)))
t0 = time.perf_counter()
kls = []
for blk in range(10):
    s.iterate(1000); kl, rl1 = s.check(); kls.append(kl)
    print(f"{(blk + 1) * 1000} iterations, {time.perf_counter() - t0:.1f} s, KL {kl:.6e}, rel-L1 {rl1:.5f}", flush=True)
W, H = s.download()
print("monotone:", all(b <= a for a, b in zip(kls, kls[1:])), "finite:", bool(np.isfinite(W).all() and np.isfinite(H).all()), "min W/H:", W.min(), H.min())
