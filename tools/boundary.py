"""Cost of the documented drop-in call (host buffers in, host buffers out) next to the resident loop: wall time of
update_div_ex and its t[] breakdown, first and repeated calls, hipGraph replay versus eager launches, for cfg3 and for
small problems."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
shapes = [(1024, 4096, 64), (4096, 350, 128), (512, 3445, 30), (4096, 65536, 256)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.replace("x", ",").split(",")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    W = ng.Matrix(rng.random((M, K), dtype=np.float32)); H = ng.Matrix(rng.random((K, N), dtype=np.float32)); X = ng.Matrix(rng.random((M, N), dtype=np.float32))
    for it, graph in ((1, 2), (200, 2), (200, 2), (200, 1), (200, 1), (200, -1), (200, 0)):
        t0 = time.perf_counter()
        r = ng.update_div_ex(W, H, X, max_iter=it, use_graph=graph)
        dt = time.perf_counter() - t0
        print(f"({M},{N},{K}) update_div host-buffer call, {it} iterations, use_graph={graph}: wall {dt * 1e3:.2f} ms; t =", {k: round(v * 1e3, 3) for k, v in r['t'].items() if v}, flush=True)
