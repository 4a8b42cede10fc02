"""What would run for a shape: the kernel family, padded extents and cut counts nmf_solver_create would choose (nmf_plan_describe: no GPU
needed), for the automatic choice and with each family forced.
    python tools/plan.py M N K [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmf_gpu_amd as ng

if len(sys.argv) < 4:
    sys.exit(__doc__)
M, N, K = (int(v) for v in sys.argv[1:4])
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for label, kw in (("automatic", {}), ("64-column kernel", {"split_kernel": -1}), ("split kernel", {"split_kernel": 1})):
    try:
        print(f"{label:>17}: {ng.plan_describe(M, N, K, batch, **kw)}")
    except ng.NmfError as e:
        print(f"{label:>17}: refused ({e})")
