"""Batched restarts: loop time of R pairs per launch as a function of the workgroup-level split counts (pick_split's batch rule)
and the whole update_div_restarts call.  python tools/restart_split_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
IT = 200
for (M, N, K, R, hs, ws) in ((4096, 350, 128, 16, (0, 1, 2, 3, 4, 6, 11), (0,)), (1024, 4096, 64, 16, (0,), (0, 1, 2, 4)), (512, 3445, 30, 16, (0,), (0, 1, 2, 4, 8)),
                             (4096, 350, 128, 4, (0, 1, 2, 3, 4, 6, 11), (0,)), (1024, 4096, 64, 4, (0,), (0, 1, 2, 4))):
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
    Ws = [np.asfortranarray(rng.random((M, K), dtype=np.float32)) for _ in range(R)]
    Hs = [np.asfortranarray(rng.random((K, N), dtype=np.float32)) for _ in range(R)]
    for nh in hs:
        for nw in ws:
            s = ng.Solver(M, N, K, batch=R, nsplit_h=nh, nsplit_w=nw)
            s.upload(None, None, X)
            for b in range(R):
                s.upload_pair(b, Ws[b], Hs[b])
            s.iterate(41); s.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); s.iterate(IT); s.sync(); best = min(best, time.perf_counter() - t0)
            d = s.describe()
            s.close()
            print(f"({M},{N},{K}) x{R} nsplit_h={nh} nsplit_w={nw}: {best / IT * 1e6:.1f} us per batch iteration = {R * IT / best:.0f} it/s aggregate = {8.0 * M * N * K * R * IT / best / 1e12:.1f} TF | {d}", flush=True)
    for _ in range(2):
        Wm, Hm = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
        Xm = ng.Matrix(X)
        t0 = time.perf_counter()
        best, kls = ng.update_div_restarts(Wm, Hm, Xm, max_iter=IT)
        dt = time.perf_counter() - t0
        print(f"({M},{N},{K}) x{R} update_div_restarts whole call: {dt * 1e3:.1f} ms = {R * IT / dt:.0f} it/s aggregate", flush=True)
