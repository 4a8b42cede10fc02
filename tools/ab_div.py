"""A/B at cfg3: default (range-guarded packed division) vs fast_divide=-1 (full IEEE sequence) vs fast_divide=1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M, N, K = 4096, 65536, 256
rng = np.random.default_rng(0)
W = np.asfortranarray(rng.random((M, K), dtype=np.float32)); H = np.asfortranarray(rng.random((K, N), dtype=np.float32))
X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
for rep in range(2):
    for fd in (0, -1, 1):
        s = ng.Solver(M, N, K, fast_divide=fd)
        s.upload(W, H, X)
        s.iterate(3); s.sync()
        r = [(s.time_piece(2, 5), s.time_piece(3, 5)) for _ in range(3)]
        print("fast_divide", fd, " H/W ms:", " ".join("%.3f/%.3f" % t for t in r), flush=True)
        s.close()
