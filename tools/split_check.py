"""Split kernel versus the CPU oracle and versus the 64-column kernel, plus iteration rates (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng, oracle
shapes = [(256, 384, 64, 10), (1024, 4096, 64, 20), (4096, 350, 128, 20), (512, 3445, 30, 20), (100, 77, 5, 10), (333, 1000, 100, 10), (2048, 2048, 128, 5)]
if len(sys.argv) > 1: shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K, iters) in shapes:
    X, W, H = oracle.gen_problem(M, N, K, seed=0)
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, iters, 25)
    out = {}
    for sk in (1, -1):
        s = ng.Solver(M, N, K, split_kernel=sk)
        s.upload(W, H, X); s.iterate(iters); s.sync()
        out[sk] = s.download()
        kl = s.check()[0]
        s.upload(W, H, X); s.iterate(20); s.sync()
        t0 = time.perf_counter(); s.iterate(200); s.sync(); dt = time.perf_counter() - t0
        print(f"({M},{N},{K}) split_kernel={sk:2d} uses_split={s.uses_split_kernel}: relF(W)={oracle.relF(out[sk][0], Wr):.2e} relF(H)={oracle.relF(out[sk][1], Hr):.2e} kl={kl:.6e}  {200 / dt:.0f} it/s ({dt / 200 * 1e6:.1f} us/it)", flush=True)
        s.close()
    print(f"    split vs 64-column kernel: relF(W)={oracle.relF(out[1][0], out[-1][0]):.2e} relF(H)={oracle.relF(out[1][1], out[-1][1]):.2e}")
