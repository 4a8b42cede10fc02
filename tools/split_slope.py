"""Fixed cost and per-superchunk cost of the split kernel: half-step time versus the reduction length (GPU box).
    python3 tools/split_slope.py [K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
def run(M, N, which):
    s = ng.Solver(M, N, K, split_kernel=1, nsplit_h=1, nsplit_w=1)
    s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)),
             np.asfortranarray(rng.random((M, N), dtype=np.float32)))
    s.time_piece(which, 5)
    t = min(s.time_piece(which, 20) for _ in range(3)) * 1e3
    s.close()
    return t
print(f"K={K}: H-step, N=4096 (256 workgroups), reduction length M:")
for M in (128, 256, 512, 1024, 2048, 4096):
    print(f"   M={M:5d} ({M // 128:3d} superchunks): {run(M, 4096, 2):7.2f} us", flush=True)
print(f"K={K}: W-step, M=4096 (256 workgroups), reduction length N:")
for N in (128, 256, 512, 1024, 2048, 4096):
    print(f"   N={N:5d} ({N // 128:3d} superchunks): {run(4096, N, 3):7.2f} us", flush=True)
