import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M,N,K = 4096,65536,256
rng = np.random.default_rng(0)
s = ng.Solver(M,N,K)
s.upload(np.asfortranarray(rng.random((M,K),dtype=np.float32)), np.asfortranarray(rng.random((K,N),dtype=np.float32)), np.asfortranarray(rng.random((M,N),dtype=np.float32)))
s.iterate(1); print("check:", s.check(), " kernel ms:", s.time_piece(6, 5))
