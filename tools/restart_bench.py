"""update_div_restarts: wall time of R restarts with lanes = 1 (one after the other) vs automatic concurrency."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
for (M, N, K, R) in ((512, 3445, 30, 16), (1024, 4096, 64, 16), (4096, 350, 128, 16), (4096, 16384, 128, 8)):
    X = np.asfortranarray(rng.random((M, N), dtype=np.float32))
    Ws = [np.asfortranarray(rng.random((M, K), dtype=np.float32)) for _ in range(R)]
    Hs = [np.asfortranarray(rng.random((K, N), dtype=np.float32)) for _ in range(R)]
    for lanes in (1, 0, 1, 0):
        Wm, Hm = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
        Xm = ng.Matrix(X)
        # the host-side preparation above leaves the GPU idle for tens of milliseconds, long enough for it to drop into a low-power
        # state: the first kernel afterwards then waits 13-21 ms for the wake-up (seen as a "drain" of the solver's first launch).
        # One tiny launch right before the timed call keeps that out of the measurement, as a warm, resident workload would.
        w = ng.Solver(64, 64, 16); w.iterate(1); w.sync(); w.close()
        t0 = time.perf_counter()
        best, kls = ng.update_div_restarts(Wm, Hm, Xm, max_iter=200, restart_lanes=lanes)
        dt = time.perf_counter() - t0
        print(f"({M},{N},{K}) x {R} restarts x 200 iterations, lanes={'auto' if lanes == 0 else lanes}: {dt * 1e3:.1f} ms = {R * 200 / dt:.0f} iterations/s, best {best}", flush=True)
