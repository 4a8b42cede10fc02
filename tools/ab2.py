import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, nmf_gpu_amd as ng
M,N,K = 4096,65536,256
rng = np.random.default_rng(0)
s = ng.Solver(M,N,K)
s.upload(np.asfortranarray(rng.random((M,K),dtype=np.float32)), np.asfortranarray(rng.random((K,N),dtype=np.float32)), np.asfortranarray(rng.random((M,N),dtype=np.float32)))
s.iterate(2); s.sync()
res = []
for rep in range(3):
    res.append((s.time_piece(2, 5), s.time_piece(3, 5)))
print("variant", os.environ.get("NMF_FUSED_VARIANT","3"), "fastdiv", os.environ.get("NMF_FAST_DIVIDE","0"), " H/W ms:", " ".join("%%.3f/%%.3f" %% r for r in res))
''' % ROOT
for var, fd in (("0","0"), ("3","0"), ("0","1"), ("0","0"), ("3","0")):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NMF_FUSED_VARIANT=var, NMF_FAST_DIVIDE=fd))
