"""BASELINE config 3, all 200 iterations, GPU (fp32) against float64 numpy on the host -- the comparison with exact-ish
arithmetic that no test can afford (~6 minutes of fp64 BLAS).  Prints the distances every 50 iterations.
    python tools/cfg3_fp64_200.py > profiles/r03_cfg3_fp64_200.log"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng, oracle
M, N, K = 4096, 65536, 256
X, W, H = oracle.gen_problem(M, N, K, seed=0)
s = ng.Solver(M, N, K)
s.upload(W, H, X)
eps = float(ng.EPS)
W64, H64, X64 = np.maximum(W.astype(np.float64), eps), np.maximum(H.astype(np.float64), eps), np.maximum(X.astype(np.float64), eps)
Z = np.empty_like(X64)
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
blk = slice(N // 2, N // 2 + 2048)
t0 = time.time()
for it in range(1, 201):
    np.matmul(W64, H64, out=Z); np.maximum(Z, eps, out=Z); np.divide(X64, Z, out=Z)
    H64 *= (W64.T @ Z) / np.maximum(W64.sum(0), eps)[:, None]
    np.matmul(W64, H64, out=Z); np.maximum(Z, eps, out=Z); np.divide(X64, Z, out=Z)
    W64 *= (Z @ H64.T) / np.maximum(H64.sum(1), eps)[None, :]
    if it % 50 == 0:
        s.iterate(50)
        Wg, Hg = s.download()
        scale = float(np.vdot(Wg.astype(np.float64), W64) / np.vdot(W64, W64)) - 1.0
        print(f"cfg3 after {it:3d} iterations, GPU vs float64 numpy: relF(W) = {rel(Wg, W64):.2e}, relF(H) = {rel(Hg, H64):.2e}, "
              f"relF(W*H block) = {rel(Wg.astype(np.float64) @ Hg[:, blk].astype(np.float64), W64 @ H64[:, blk]):.2e}, scale of W {scale:+.1e}  [{time.time() - t0:.0f} s]", flush=True)
s.close()
