"""update_div_restarts, 16 x 200 iterations: eager launches (the automatic choice for runs this short) against hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
for (M, N, K, R) in ((512, 3445, 30, 16), (1024, 4096, 64, 16), (4096, 350, 128, 16)):
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
    Ws = [np.asfortranarray(rng.random((M, K), dtype=np.float32)) for _ in range(R)]
    Hs = [np.asfortranarray(rng.random((K, N), dtype=np.float32)) for _ in range(R)]
    for graph in (-1, 1, -1, 1):
        Wm, Hm = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
        Xm = ng.Matrix(X)
        w = ng.Solver(64, 64, 16); w.iterate(1); w.sync(); w.close()
        t0 = time.perf_counter()
        best, kls = ng.update_div_restarts(Wm, Hm, Xm, max_iter=200, use_graph=graph)
        dt = time.perf_counter() - t0
        print(f"({M},{N},{K}) x {R} x 200, use_graph={graph}: {dt * 1e3:.1f} ms", flush=True)
