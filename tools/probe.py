import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
M,N,K = 4096,65536,256
rng = np.random.default_rng(0)
s = ng.Solver(M,N,K)
s.upload(np.asfortranarray(rng.random((M,K),dtype=np.float32)), np.asfortranarray(rng.random((K,N),dtype=np.float32)), np.asfortranarray(rng.random((M,N),dtype=np.float32)))
s.iterate(1); s.sync()
kinds = {0:"v_add_f32", 4:"v_accvgpr_read", 3:"global_load_dword", 6:"global_load_dwordx4", 1:"ds_read_b32", 5:"ds_write_b32", 7:"ds_write_b128", 2:"s_add_u32"}
for kind in kinds:
    row = []
    for nv in (0, 1, 2, 4):
        t = min(s.time_piece(1000 + 10*kind + nv, 3) for _ in range(2))
        row.append(t*1e-3*2.4e9/(2000*64))
    print(f"probe {kinds[kind]:20s} cycles/MFMA @2.4GHz with 0/1/2/4 per MFMA: " + " ".join(f"{c:6.1f}" for c in row))
for mode, name in ((0, "partner idle"), (1, "partner VALU"), (2, "partner ds_read")):
    t = min(s.time_piece(2000 + mode, 3) for _ in range(2))
    print(f"partner-probe {name:16s}: {t:.3f} ms")
print("stamp build ms:", s.time_piece(3000, 1))   # per-segment cycles go to stderr
s.time_piece(4000, 1)                              # divide census goes to stderr
