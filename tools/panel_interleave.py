"""Round-4 VERDICT next 6: at K <= 48 and N >= 32768 an iteration is HBM-bound (K = 16: 4.8 TB/s) and reads X twice, once per half-step.
Experiment: run the iteration PER COLUMN PANEL -- H-step of panel p, then the W-step's partial product over the same columns while the
panel of X is still in the 256 MiB Infinity Cache -- and see whether the second read gets cheaper.  The panels are the library's own
column shards driven through the half-step protocol (update_h / w_partial / w_apply, the sharded run's pieces) on ONE stream, the P
partial buffers summed in order by torch, the whole iteration captured in a graph and replayed.  P = 1 is the same protocol on the
whole problem (the control); `production` is the solver's own hipGraph iteration.
    python tools/panel_interleave.py [M N K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nmf_gpu_amd as ng

shapes = [(4096, 65536, 16), (4096, 65536, 32), (4096, 65536, 48), (4096, 131072, 16)]
if len(sys.argv) > 1:
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
rng = np.random.default_rng(0)
dev = "cuda:0"
for (M, N, K) in shapes:
    W = np.asfortranarray(rng.random((M, K), dtype=np.float32)); H = np.asfortranarray(rng.random((K, N), dtype=np.float32))
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
    s = ng.Solver(M, N, K)
    s.upload(W, H, X); s.iterate(41); s.sync()
    t0 = time.perf_counter(); s.iterate(128); s.sync(); prod = (time.perf_counter() - t0) / 128
    desc = s.describe()
    s.close()
    print(f"({M},{N},{K}) production: {prod * 1e3:.4f} ms/iteration   [{desc}]", flush=True)
    ref_W = None
    for P in (1, 2, 4, 8, 16):
        st = torch.cuda.Stream(device=dev)
        cols = ng.column_shards(N, P)
        sol = [ng.Solver(M, c1 - c0, K, device=0, stream=st.cuda_stream, use_graph=False) for (c0, c1) in cols]
        cnt = sol[0].partial_buffer()[1]
        with torch.cuda.stream(st):
            bufs = torch.zeros((P, cnt), dtype=torch.float32, device=dev)
        st.synchronize()
        for p, (c0, c1) in enumerate(cols):
            sol[p].set_partial_buffer(bufs[p].data_ptr(), cnt)
            sol[p].upload(W, np.asfortranarray(H[:, c0:c1]), np.asfortranarray(X[:, c0:c1]))

        def iteration():
            for p in range(P):
                sol[p].update_h()
                sol[p].w_partial()
            if P > 1:
                tot = bufs[0].clone()
                for p in range(1, P):
                    tot += bufs[p]
                bufs[:] = tot
            for p in range(P):
                sol[p].w_apply()

        with torch.cuda.stream(st):
            for _ in range(3):
                iteration()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            iteration()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        reps = 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        Wp, _ = sol[0].download()
        if ref_W is None:
            ref_W = Wp
        err = float(np.linalg.norm(Wp - ref_W) / np.linalg.norm(ref_W))
        panel_mib = 4.0 * M * (N / P) / 1048576.0
        print(f"    P = {P:2d} panels of {panel_mib:7.1f} MiB of X: {ms:.4f} ms/iteration ({100 * (prod * 1e3 / ms - 1):+.1f} % against production), "
              f"relF(W) against P = 1 after {3 + 1 + 5 + reps} iterations {err:.1e}", flush=True)
        for x in sol:
            x.close()
        del g, bufs
