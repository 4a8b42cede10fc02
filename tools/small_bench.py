"""The three small shapes (BASELINE config 2, the reference's gold shape matrix_export.py:4-7, the paper's example): iterations/s
with W/H/X resident, hipGraph replay and eager launches; first the 200-iteration run of the reference (cuda/nmf.cu:10) started
right after a 41-iteration warm-up (every graph level captured), then a sustained run of 20 000 iterations (the GPU's clocks take
longer than a 7 ms run to settle: the sustained rate is the higher one)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
for (M, N, K) in ((1024, 4096, 64), (4096, 350, 128), (512, 3445, 30)):
    for graph in (True, False):
        s = ng.Solver(M, N, K, use_graph=graph)
        s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)), np.asfortranarray(rng.random((M, N), dtype=np.float32)))
        s.iterate(41); s.sync()          # 32 + 8 + 1: every graph level captured and instantiated before the timed region
        t0 = time.perf_counter(); s.iterate(200); s.sync(); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); s.iterate(20000); s.sync(); dl = time.perf_counter() - t0
        print(f"({M},{N},{K}) graph={graph}: 200 iterations {dt * 1e3:.2f} ms = {200 / dt:.0f} it/s, {8 * M * N * K * 200 / dt / 1e12:.2f} TF; "
              f"sustained (20000 iterations) {20000 / dl:.0f} it/s, {8 * M * N * K * 20000 / dl / 1e12:.2f} TF; kernels H/W {s.time_piece(2, 20) * 1e3:.1f}/{s.time_piece(3, 20) * 1e3:.1f} us", flush=True)
        s.close()
