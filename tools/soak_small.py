"""Long runs on the split kernel: cfg2, the gold shape and the paper's shape, 50 000 iterations each with a KL check every
5000; KL must not increase, factors must stay finite; and the final state must be bit-identical between two such runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 4096, 64), (4096, 350, 128), (512, 3445, 30)]
for (M, N, K) in shapes:
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32)); W = np.asfortranarray(rng.random((M, K), dtype=np.float32)); H = np.asfortranarray(rng.random((K, N), dtype=np.float32))
    finals = []
    for rep in range(2):
        s = ng.Solver(M, N, K)
        s.upload(W, H, X)
        t0 = time.perf_counter()
        r = s.run(1e-30, 50000, 5000)
        dt = time.perf_counter() - t0
        Wg, Hg = s.download(); s.close()
        kl = np.asarray(r["kl"])
        ok = bool(np.all(np.diff(kl) <= 0) and np.isfinite(Wg).all() and np.isfinite(Hg).all())
        finals.append((Wg, Hg))
        print(f"({M},{N},{K}) run {rep}: {r['iterations']} iterations in {dt:.2f} s = {r['iterations'] / dt:.0f} it/s incl. 11 checks; KL {kl[0]:.4e} -> {kl[-1]:.6e}, non-increasing and finite: {ok}", flush=True)
    print("    two runs bit-identical:", bool(np.array_equal(finals[0][0], finals[1][0]) and np.array_equal(finals[0][1], finals[1][1])), flush=True)
