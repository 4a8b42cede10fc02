"""CPU oracle for the update_div hot path -- TEST INFRASTRUCTURE ONLY.

ctypes bindings over ``oracle/libnmf_oracle.so`` (built from ``nmf_oracle.c`` by
``oracle/Makefile``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product (``nmf-gpu_amd/``) never does.

Parity status: ``refcompat`` mode is pinned by the reference's golden outputs
(``tests/golden/Wtest.bin`` / ``Htest.bin``); ``spec`` mode shares all arithmetic with it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
EPS = np.float32(2.2204e-16)
MODE_SPEC, MODE_REFCOMPAT = 0, 1
_lib = None
_lib_native = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(native: bool = False) -> str:
    """(Re)build the oracle shared library with gcc; returns its path."""
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, "libnmf_oracle_native.so" if native else "libnmf_oracle.so")


def cpu_budget() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup cpu quota).  The GPU box shows all
    host CPUs but caps the container by quota; an OpenMP team sized to the former thrashes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))
    # last resort when the quota is not visible: a 1-GPU box grants about 16 CPUs
    n = min(n, int(os.environ.get("NMF_ORACLE_MAX_THREADS", "16")))
    return max(1, n)


def _bind(path: str):
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))
    lib = C.CDLL(path)
    lib.oracle_set_num_threads.restype = None
    lib.oracle_set_num_threads.argtypes = [C.c_int]
    lib.oracle_set_num_threads(cpu_budget())
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_update_div.restype = C.c_int
    lib.oracle_update_div.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int,
                                      C.POINTER(C.c_int)]
    for name in ("oracle_sgemm_nn", "oracle_sgemm_tn", "oracle_sgemm_nt"):
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p]
    lib.oracle_set_epsilon.restype = None
    lib.oracle_set_epsilon.argtypes = [_f32p, C.c_size_t]
    for name in ("oracle_sum_cols", "oracle_sum_rows", "oracle_sum_cols_refcompat"):
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
    lib.oracle_kl_div.restype = C.c_double
    lib.oracle_kl_div.argtypes = [_f32p, _f32p, C.c_size_t]
    lib.oracle_rel_l1.restype = C.c_double
    lib.oracle_rel_l1.argtypes = [_f32p, _f32p, C.c_size_t]
    for name in ("oracle_update_h", "oracle_update_w"):
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int]
    lib.oracle_fast_update_div.restype = C.c_int
    lib.oracle_fast_update_div.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.oracle_rng_seed.restype = None
    lib.oracle_rng_seed.argtypes = [C.c_void_p, C.c_uint32]
    lib.oracle_rng_fill_f32.restype = None
    lib.oracle_rng_fill_f32.argtypes = [C.c_void_p, _f32p, C.c_size_t]
    lib.oracle_read_bin.restype = C.c_int
    lib.oracle_read_bin.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.POINTER(C.c_float))]
    lib.oracle_write_bin.restype = C.c_int
    lib.oracle_write_bin.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _f32p]
    return lib


def lib(native: bool = False):
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            _lib_native = _bind(build(native=True))
        return _lib_native
    if _lib is None:
        path = os.path.join(_HERE, "libnmf_oracle.so")
        srcs = [os.path.join(_HERE, f) for f in ("nmf_oracle.c", "nmf_oracle_fast.c", "nmf_oracle.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
            build()
        _lib = _bind(path)
    return _lib


# ----------------------------------------------------------------------------- inputs
class Rng:
    """numpy.random.seed(seed) / rand(n).astype(float32) restated (matrix_export.py:4-7)."""

    def __init__(self, seed: int = 0):
        self._state = (C.c_uint32 * 625)()
        lib().oracle_rng_seed(C.addressof(self._state), seed)

    def rand_f32(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        lib().oracle_rng_fill_f32(C.addressof(self._state), out, n)
        return out


def gen_problem(M: int, N: int, K: int, seed: int = 0):
    """X (M x N), W (M x K), H (K x N), drawn X -> W -> H from one MT19937 stream like
    matrix_export.py:4-7.  Returned as Fortran-ordered (column-major) float32 arrays whose
    flat buffers are byte-identical to the payload of X.bin / W.bin / H.bin."""
    g = Rng(seed)
    X = g.rand_f32(M * N).reshape((M, N), order="F")
    W = g.rand_f32(M * K).reshape((M, K), order="F")
    H = g.rand_f32(K * N).reshape((K, N), order="F")
    return X, W, H


def _flat(a: np.ndarray) -> np.ndarray:
    """column-major flat view (no copy for F-ordered input)."""
    a = np.asarray(a, dtype=np.float32)
    return np.ascontiguousarray(a.reshape(-1, order="F"))


# ------------------------------------------------------------------------------- loop
def update_div(W, H, X, thresh: float = 0.0, max_iter: int = 200, iter_check: int = 25,
               mode: int = MODE_SPEC, native: bool = False):
    """Run the oracle loop.  Returns (W, H, iters, kl_trace) with W,H F-ordered copies."""
    M, K = W.shape
    K2, N = H.shape
    assert K == K2 and X.shape == (M, N)
    w, h, x = _flat(W).copy(), _flat(H).copy(), _flat(X)
    cap = 2 + (max_iter // iter_check if iter_check > 0 else 0)
    kl = (C.c_double * cap)()
    nkl = C.c_int(0)
    it = lib(native).oracle_update_div(w, h, x, M, N, K, thresh, max_iter, iter_check, mode, kl, cap,
                                       C.byref(nkl))
    return (w.reshape((M, K), order="F"), h.reshape((K, N), order="F"), it,
            np.array(kl[: min(nkl.value, cap)]))


def update_div_fast(W, H, X, iters: int = 1, native: bool = False):
    """`iters` spec-mode iterations through the register-tiled baseline kernels (nmf_oracle_fast.c): what bench.py
    times as the CPU baseline.  Returns (W, H) as F-ordered copies."""
    M, K = W.shape
    N = H.shape[1]
    w, h, x = _flat(W).copy(), _flat(H).copy(), _flat(X)
    if lib(native).oracle_fast_update_div(w, h, x, M, N, K, iters) != 0:
        raise MemoryError("oracle_fast_update_div: scratch allocation failed")
    return w.reshape((M, K), order="F"), h.reshape((K, N), order="F")


def update_h(W, H, X, mode: int = MODE_SPEC):
    """One H half-step on clamped inputs; returns new H (F-ordered)."""
    M, K = W.shape
    N = H.shape[1]
    w, h, x = _flat(W).copy(), _flat(H).copy(), _flat(X)
    Z = np.empty(M * N, np.float32); WtZ = np.empty(K * N, np.float32); s = np.empty(K, np.float32)
    lib().oracle_update_h(w, h, x, M, N, K, Z, WtZ, s, mode)
    return h.reshape((K, N), order="F")


def update_w(W, H, X, mode: int = MODE_SPEC):
    """One W half-step on clamped inputs; returns new W (F-ordered)."""
    M, K = W.shape
    N = H.shape[1]
    w, h, x = _flat(W).copy(), _flat(H).copy(), _flat(X)
    Z = np.empty(M * N, np.float32); ZHt = np.empty(M * K, np.float32); s = np.empty(K, np.float32)
    lib().oracle_update_w(w, h, x, M, N, K, Z, ZHt, s, mode)
    return w.reshape((M, K), order="F")


def clamp(a):
    out = _flat(a).copy()
    lib().oracle_set_epsilon(out, out.size)
    return out.reshape(a.shape, order="F")


def sgemm(kind: str, A, B):
    """kind in {'nn','tn','nt'}: A*B, A'*B, A*B' (column-major fp32)."""
    a, b = _flat(A), _flat(B)
    if kind == "nn":
        m, k = A.shape; n = B.shape[1]
    elif kind == "tn":
        k, m = A.shape; n = B.shape[1]
    else:
        m, k = A.shape; n = B.shape[0]
    c = np.empty(m * n, np.float32)
    getattr(lib(), "oracle_sgemm_" + kind)(m, n, k, a, b, c)
    return c.reshape((m, n), order="F")


def sum_cols(A, refcompat: bool = False):
    out = np.empty(A.shape[1], np.float32)
    f = lib().oracle_sum_cols_refcompat if refcompat else lib().oracle_sum_cols
    f(_flat(A), A.shape[0], A.shape[1], out)
    return out


def sum_rows(A):
    out = np.empty(A.shape[0], np.float32)
    lib().oracle_sum_rows(_flat(A), A.shape[0], A.shape[1], out)
    return out


def kl_div(X, Y) -> float:
    return float(lib().oracle_kl_div(_flat(X), _flat(Y), X.size))


def rel_l1(X, Y) -> float:
    return float(lib().oracle_rel_l1(_flat(X), _flat(Y), X.size))


def read_bin(path: str) -> np.ndarray:
    r, c = C.c_uint32(), C.c_uint32()
    p = C.POINTER(C.c_float)()
    rc = lib().oracle_read_bin(path.encode(), C.byref(r), C.byref(c), C.byref(p))
    if rc != 0:
        raise IOError(f"oracle_read_bin({path}) -> {rc}")
    n = r.value * c.value
    arr = np.ctypeslib.as_array(p, shape=(n,)).copy()
    C.CDLL(None).free(p)
    return arr.reshape((r.value, c.value), order="F")


def write_bin(path: str, A) -> None:
    rc = lib().oracle_write_bin(path.encode(), A.shape[0], A.shape[1], _flat(A))
    if rc != 0:
        raise IOError(f"oracle_write_bin({path}) -> {rc}")


def relF(a, b) -> float:
    """rel-Frobenius ||a-b|| / ||b|| in fp64 (the parity metric)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def num_threads(native: bool = False) -> int:
    return int(lib(native).oracle_num_threads())
