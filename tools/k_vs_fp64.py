"""GPU vs oracle vs float64 numpy after 200 iterations at K = 700 (wave-pair kernel), K = 576 (64-column kernel) and K = 512 (control):
is the 1e-5 .. 2e-5 between GPU and oracle at K > M summation-order noise of both, or something of the new kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tools/ -> repo root
import numpy as np, nmf_gpu_amd as ng, oracle
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for (M, N, K) in ((512, 2048, 700), (1024, 2048, 700), (512, 2048, 576), (512, 2048, 512), (512, 2048, 448)):
    X, W, H = oracle.gen_problem(M, N, K, seed=K)
    eps, iters = float(ng.EPS), 200
    W64, H64, X64 = (np.ascontiguousarray(np.maximum(a.astype(np.float64), eps)) for a in (W, H, X))
    Z = np.empty_like(X64)
    t0 = time.time()
    for _ in range(iters):
        np.matmul(W64, H64, out=Z); np.maximum(Z, eps, out=Z); np.divide(X64, Z, out=Z)
        H64 *= (W64.T @ Z) / np.maximum(W64.sum(0), eps)[:, None]
        np.matmul(W64, H64, out=Z); np.maximum(Z, eps, out=Z); np.divide(X64, Z, out=Z)
        W64 *= (Z @ H64.T) / np.maximum(H64.sum(1), eps)[None, :]
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=iters, use_graph=1)
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, iters, 25)
    s = ng.Solver(M, N, K); d = s.describe(); s.close()
    print(f"({M},{N},{K}) [{d.split(' ')[0]}] fp64 {time.time() - t0:.0f} s: GPU vs oracle W {rel(Wm.mat, Wr.astype(np.float64)):.2e} H {rel(Hm.mat, Hr.astype(np.float64)):.2e} | "
          f"GPU vs fp64 W {rel(Wm.mat, W64):.2e} H {rel(Hm.mat, H64):.2e} WH {rel(Wm.mat.astype(np.float64) @ Hm.mat.astype(np.float64), W64 @ H64):.2e} | "
          f"oracle vs fp64 W {rel(Wr, W64):.2e} H {rel(Hr, H64):.2e} WH {rel(Wr.astype(np.float64) @ Hr.astype(np.float64), W64 @ H64):.2e}", flush=True)
