import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M,N,K = 4096,65536,256
rng = np.random.default_rng(0)
W=ng.Matrix(rng.random((M,K),dtype=np.float32)); H=ng.Matrix(rng.random((K,N),dtype=np.float32)); X=ng.Matrix(rng.random((M,N),dtype=np.float32))
for it in (1, 200):
    t0=time.perf_counter()
    r = ng.update_div_ex(W,H,X,max_iter=it)
    dt=time.perf_counter()-t0
    print(f"update_div host-buffer call, {it} iterations: wall {dt:.3f} s; t =", {k: round(v,4) for k,v in r['t'].items() if v})
