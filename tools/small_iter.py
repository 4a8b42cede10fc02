"""One small shape, one launch mode, for rocprofv3 --kernel-trace:
    python3 tools/small_iter.py M N K graph(0|1) [iters] [split_kernel] [nsplit_h] [nsplit_w] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M, N, K, graph = (int(v) for v in sys.argv[1:5])
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
iters, sk, nh, nw, batch = arg(5, 64), arg(6, 0), arg(7, 0), arg(8, 0), arg(9, 1)
rng = np.random.default_rng(0)
s = ng.Solver(M, N, K, use_graph=bool(graph), split_kernel=sk, nsplit_h=nh, nsplit_w=nw, batch=batch)
s.upload(None, None, np.asfortranarray(rng.random((M, N), dtype=np.float32)))
for b in range(batch):
    s.upload_pair(b, np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)))
s.iterate(41); s.sync()
t0 = time.perf_counter(); s.iterate(iters); s.sync(); dt = time.perf_counter() - t0
print(f"({M},{N},{K}) graph={graph} split={s.uses_split_kernel} nsplit=({nh},{nw}) batch={batch}: {iters} iterations {dt * 1e3:.2f} ms = {iters * batch / dt:.0f} it/s "
      f"({dt / iters * 1e6:.1f} us per launch set, {8 * M * N * K * iters * batch / dt / 1e12:.2f} TF)")
s.close()
