import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M,N,K = 4096,65536,256
rng = np.random.default_rng(0)
W=np.asfortranarray(rng.random((M,K),dtype=np.float32)); H=np.asfortranarray(rng.random((K,N),dtype=np.float32)); X=np.asfortranarray(rng.random((M,N),dtype=np.float32))
for nsw in (4, 8, 16, 24, 32):
    s = ng.Solver(M,N,K, nsplit_w=nsw)
    s.upload(W,H,X); s.iterate(2); s.sync()
    import time
    t0=time.perf_counter(); s.iterate(20); s.sync(); dt=(time.perf_counter()-t0)/20*1e3
    print(f"nsplit_w={nsw:2d}: W-step kernel {s.time_piece(3,5):.3f} ms, iteration {dt:.3f} ms")
    s.close()
