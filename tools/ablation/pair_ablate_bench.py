"""H-step / W-step time of the wave-pair kernel under each ablation build (ab/libnmf_pair_ablate_<n>.so): where do the ~1250 cycles per
16-row chunk go that the MFMAs do not account for?  One subprocess per build (the library is bound at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {0: "as shipped", 1: "no quotient", 2: "no LDS exchange of the halves of S (and its barrier)", 3: "no end-of-chunk barrier", 4: "no nops behind product 1",
         5: "no ds_writes of the next image", 6: "no global loads in the loop"}
code = """
import sys, os
sys.path.insert(0, %r)
import numpy as np, nmf_gpu_amd as ng
rng = np.random.default_rng(0)
for (M, N, K) in ((4096, 32768, 640), (4096, 32768, 1024)):
    s = ng.Solver(M, N, K, use_graph=False)
    s.upload(np.asfortranarray(rng.random((M, K), dtype=np.float32)), np.asfortranarray(rng.random((K, N), dtype=np.float32)), np.asfortranarray(rng.random((M, N), dtype=np.float32)))
    s.iterate(2); s.sync()
    h = min(s.time_piece(2, 5) for _ in range(3)); w = min(s.time_piece(3, 5) for _ in range(3))
    f = 4.0 * M * N * K
    print(f"  K={K}: H-step {h:.3f} ms = {f / h / 1e9:.1f} TF   W-step {w:.3f} ms = {f / w / 1e9:.1f} TF", flush=True)
    s.close()
""" % ROOT
for n in (0, 1, 2, 3, 4, 5, 6, 0):
    print(f"ablation {n}: {NAMES[n]}", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NMF_LIB_PATH=os.path.join(ROOT, "scratch_ab", f"libnmf_pair_ablate_{n}.so")), check=False)
