"""Same-box A/B of nmf_opts.fast_divide = 0 (range-guarded correctly rounded quotient, the default) against 1 (the unguarded six-instruction
quotient): ms per hipGraph-replayed iteration, the two settings interleaved, best of `rounds` per setting.  Decides whether the DIV = 1
instantiations earn their half of the kernel count (round-4 VERDICT next 5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
shapes = [(4096, 65536, 256, 64), (4096, 65536, 64, 128), (4096, 65536, 16, 256), (1024, 4096, 64, 4000), (4096, 350, 128, 4000), (512, 3445, 30, 4000), (4096, 65536, 640, 24)]
rounds = 4
rng = np.random.default_rng(0)
for (M, N, K, iters) in shapes:
    W = np.asfortranarray(rng.random((M, K), dtype=np.float32)); H = np.asfortranarray(rng.random((K, N), dtype=np.float32))
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
    solvers = {}
    for fd in (0, 1):
        s = ng.Solver(M, N, K, fast_divide=fd)
        s.upload(W, H, X)
        s.iterate(41); s.sync()
        solvers[fd] = s
    best = {0: 1e9, 1: 1e9}
    for r in range(rounds):
        for fd in (0, 1) if r % 2 == 0 else (1, 0):
            s = solvers[fd]
            t0 = time.perf_counter(); s.iterate(iters); s.sync(); dt = (time.perf_counter() - t0) / iters
            best[fd] = min(best[fd], dt)
    print(f"({M},{N},{K}): fast_divide=0 {best[0] * 1e3:.4f} ms/iteration, fast_divide=1 {best[1] * 1e3:.4f} ms/iteration: {100 * (best[0] / best[1] - 1):+.2f} % for the unguarded quotient   [{solvers[0].describe()}]", flush=True)
    for s in solvers.values():
        s.close()
