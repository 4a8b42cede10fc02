import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
K = 256
for (M, N) in ((2048, 65536), (4096, 65536), (8192, 65536), (4096, 32768), (4096, 131072)):
    s = ng.Solver(M, N, K)
    s.upload(np.asfortranarray(rng.random((M,K),dtype=np.float32)), np.asfortranarray(rng.random((K,N),dtype=np.float32)), np.asfortranarray(rng.random((M,N),dtype=np.float32)))
    s.iterate(2); s.sync()
    h = min(s.time_piece(2, 5) for _ in range(2)); w = min(s.time_piece(3, 5) for _ in range(2))
    fl = 4.0*M*N*K
    print(f"M={M:5d} N={N:6d}: H {h:.3f} ms ({fl/h/1e9:.1f} TF)  W {w:.3f} ms ({fl/w/1e9:.1f} TF)")
    s.close()
