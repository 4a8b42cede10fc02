"""Fused half-step times and whole-iteration rate for a list of (M, N, K) shapes (W/H/X resident, hipGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
shapes = [(8192, 16384, 512), (8192, 131072, 512), (4096, 65536, 128), (4096, 65536, 64), (4096, 262144, 256), (4096, 32768, 256), (4096, 65536, 384)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.replace("x", ",").split(",")) for a in sys.argv[1:]]
rng = np.random.default_rng(0)
for (M, N, K) in shapes:
    s = ng.Solver(M, N, K)
    s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)),
             np.asfortranarray(rng.random((M, N), dtype=np.float32)))
    s.iterate(41); s.sync()          # every graph level captured, clocks settled
    t0 = time.perf_counter(); s.iterate(64); s.sync(); dt = (time.perf_counter() - t0) / 64
    f = 4.0 * M * N * K
    if s.path == ng.PATH_FUSED:
        h, w = s.time_piece(2, 10), s.time_piece(3, 10)
        print(f"({M},{N},{K}): iteration {dt * 1e3:.3f} ms = {2 * f / dt / 1e12:.1f} TFLOP/s effective; H-step {h:.3f} ms = {f / h / 1e9:.1f} TF, W-step {w:.3f} ms = {f / w / 1e9:.1f} TF", flush=True)
    else:   # operator path (K > 512): no fused half-step kernels to time
        print(f"({M},{N},{K}): iteration {dt * 1e3:.3f} ms = {2 * f / dt / 1e12:.1f} TFLOP/s effective (operator path)", flush=True)
    s.close()
