"""Wave-pair kernel (512 < K <= 1024) versus the oracle, versus the operator path, and its rate (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng, oracle
for (M, N, K, iters) in [(256, 384, 640, 3), (200, 130, 1000, 3), (512, 4096, 1024, 3), (96, 520, 768, 2), (320, 64, 900, 2)]:
    X, W, H = oracle.gen_problem(M, N, K, seed=0)
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, iters, 25)
    for path in (ng.PATH_FUSED, ng.PATH_UNFUSED):
        s = ng.Solver(M, N, K, path=path)
        s.upload(W, H, X); s.iterate(iters); Wg, Hg = s.download()
        kl = s.check()[0]
        y = np.maximum(Wg.astype(np.float64) @ Hg.astype(np.float64), 2.2204e-16); x = np.maximum(X, 2.2204e-16).astype(np.float64)
        klr = float((x * (np.log(x) - np.log(y)) - x + y).sum())
        print(f"({M},{N},{K}) {s.describe()}: relF(W)={oracle.relF(Wg, Wr):.2e} relF(H)={oracle.relF(Hg, Hr):.2e} kl rel err {abs(kl - klr) / klr:.1e}", flush=True)
        s.close()
