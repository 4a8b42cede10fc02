"""nmf-gpu_amd: MI355X-native (gfx950) implementation of the update_div hot path of
recoord/nmf-gpu (KL-divergence multiplicative-update NMF, X ~ W*H).

The product is ``libnmf_mi355x.so`` (HIP kernels + C++ host loop, C ABI in
``include/nmf_mi355x.h``).  This Python layer is a thin ctypes mirror of that ABI with the
reference's names (``update_div``, ``read_matrix``, ``matrix_multiply`` ...).  There is no CPU
fallback: importing :mod:`nmf_gpu_amd.api` raises if the HIP library has not been built.
"""
from .api import *  # noqa: F401,F403
from .api import __all__  # noqa: F401
from .sharded import ShardedLoop, GpuShard, column_shards, worth_sharding, negotiate_comm  # noqa: F401,E402
