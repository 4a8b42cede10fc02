"""Single small problems: iteration rate as a function of the workgroup-level split counts (pick_split's single-pair rule).
python tools/single_split_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
IT = 4000
for (M, N, K, hs, ws) in ((1024, 4096, 64, (0, 2), (0, 2, 8, 16)), (4096, 350, 128, (0, 4, 8, 16, 32), (0, 2)), (512, 3445, 30, (0, 2), (0, 4, 14, 27))):
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
    W = np.asfortranarray(rng.random((M, K), dtype=np.float32))
    H = np.asfortranarray(rng.random((K, N), dtype=np.float32))
    for nh in hs:
        for nw in ws:
            if nh and nw:
                continue
            s = ng.Solver(M, N, K, nsplit_h=nh, nsplit_w=nw)
            s.upload(W, H, X)
            s.iterate(41); s.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); s.iterate(IT); s.sync(); best = min(best, time.perf_counter() - t0)
            print(f"({M},{N},{K}) nsplit_h={nh} nsplit_w={nw}: {best / IT * 1e6:.2f} us per iteration = {IT / best:.0f} it/s | {s.describe()}", flush=True)
            s.close()
