"""ctypes mirror of include/nmf_mi355x.h with the reference's operator names.

Reference interface mirrored (paths in the reference repository):
  * ``update_div(W, H, X, CONVERGE_THRESH, max_iter, t, verbose)``  README.md:40-54
  * ``read_matrix`` / ``write_matrix``                               cuda/nmf.cu:188-259
  * ``class Matrix`` and the operators of cuda/matrix.cuh:18-52
  * ``run_async`` (as :class:`Solver`)                              cuda/nmf.cu:76-116

Nothing here computes: every call goes through the C ABI into the HIP library.  If the
library is missing this module raises at import -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

__all__ = [
    "EPS", "Matrix", "Solver", "NmfError", "update_div", "update_div_ex", "update_div_restarts", "read_matrix", "write_matrix",
    "matrix_multiply", "matrix_multiply_AtB", "matrix_multiply_ABt", "element_multiply", "element_divide",
    "row_divide", "col_divide", "set_epsilon", "sum_cols", "sum_rows", "kl_divergence", "diff_norm",
    "Comm", "device_count", "device_name", "lib", "LIB_PATH", "PATH_AUTO", "PATH_FUSED", "PATH_UNFUSED",
    "T_NAMES", "declared_symbols", "comm_library_info", "comm_capture_refusals", "record_kernels", "last_kernel", "plan_describe",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMF_LIB_PATH") or os.path.join(_HERE, "libnmf_mi355x.so")   # override: A/B builds
EPS = np.float32(2.2204e-16)
PATH_AUTO, PATH_FUSED, PATH_UNFUSED = 0, 1, 2
T_NAMES = ["total", "h2d", "h_step", "w_step", "sums", "apply", "check", "allreduce", "d2h", "setup"]
T_H_STEP, T_W_STEP, T_SUMS, T_APPLY, T_CHECK = 2, 3, 4, 5, 6
NMF_MAX_KL = 64


class NmfError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"nmf status {status}: {msg}")
        self.status = status


class _matrix(C.Structure):
    """``matrix`` of include/nmf_mi355x.h (README.md:31-36)."""
    _fields_ = [("mat", C.POINTER(C.c_float)), ("mat_d", C.POINTER(C.c_float)), ("dim", C.c_int * 2)]


class _opts(C.Structure):
    _fields_ = [("converge_thresh", C.c_float), ("max_iter", C.c_int), ("iter_check", C.c_int),
                ("verbose", C.c_int), ("path", C.c_int), ("use_graph", C.c_int), ("device", C.c_int),
                ("stream", C.c_void_p), ("comm", C.c_void_p), ("nsplit_h", C.c_int), ("nsplit_w", C.c_int),
                ("fast_divide", C.c_int), ("restart_lanes", C.c_int), ("split_kernel", C.c_int),
                ("n_devices", C.c_int), ("devices", C.POINTER(C.c_int)), ("emulate_shards", C.c_int)]


class _result(C.Structure):
    _fields_ = [("iterations", C.c_int), ("n_kl", C.c_int), ("kl", C.c_double * NMF_MAX_KL),
                ("rel_l1", C.c_double), ("path_used", C.c_int), ("t", C.c_double * 10),
                ("n_shards", C.c_int), ("w_replicas_identical", C.c_int)]


# every symbol include/nmf_mi355x.h declares: (name, restype, argtypes)
_f32p = C.POINTER(C.c_float)
_SIGS = [
    ("update_div", None, [_matrix, _matrix, _matrix, C.c_float, C.c_int, C.POINTER(C.c_double), C.c_int]),
    ("update_div_ex", C.c_int, [_matrix, _matrix, _matrix, C.POINTER(_opts), C.POINTER(_result)]),
    ("update_div_restarts", C.c_int, [C.POINTER(_matrix), C.POINTER(_matrix), C.c_int, _matrix, C.POINTER(_opts), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("nmf_default_opts", None, [C.POINTER(_opts)]),
    ("nmf_status_string", C.c_char_p, [C.c_int]),
    ("nmf_last_error", C.c_char_p, []),
    ("nmf_create_matrix", C.c_int, [C.POINTER(_matrix), C.c_int, C.c_int, C.c_float]),
    ("nmf_destroy_matrix", None, [C.POINTER(_matrix)]),
    ("nmf_read_matrix", C.c_int, [C.POINTER(_matrix), C.c_char_p]),
    ("nmf_write_matrix", C.c_int, [_matrix, C.c_char_p]),
    ("nmf_matrix_alloc_device", C.c_int, [C.POINTER(_matrix), C.c_int, C.c_int]),
    ("nmf_matrix_to_device", C.c_int, [C.POINTER(_matrix)]),
    ("nmf_matrix_from_device", C.c_int, [C.POINTER(_matrix)]),
    ("nmf_matrix_free_device", C.c_int, [C.POINTER(_matrix)]),
    ("nmf_matrix_multiply", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_matrix_multiply_AtB", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_matrix_multiply_ABt", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_element_multiply", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_element_divide", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_row_divide", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_col_divide", C.c_int, [_matrix, _matrix, _matrix, C.c_void_p]),
    ("nmf_set_epsilon", C.c_int, [_matrix, C.c_void_p]),
    ("nmf_sum_cols", C.c_int, [_matrix, _matrix, C.c_void_p]),
    ("nmf_sum_rows", C.c_int, [_matrix, _matrix, C.c_void_p]),
    ("nmf_kl_divergence", C.c_int, [_matrix, _matrix, C.POINTER(C.c_double), C.c_void_p]),
    ("nmf_diff_norm", C.c_int, [_matrix, _matrix, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    ("nmf_solver_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(_opts)]),
    ("nmf_solver_destroy", None, [C.c_void_p]),
    ("nmf_solver_upload", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nmf_solver_upload_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nmf_solver_download", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nmf_solver_iterate", C.c_int, [C.c_void_p, C.c_int]),
    ("nmf_solver_prepare", C.c_int, [C.c_void_p, C.c_int]),
    ("nmf_solver_iterate_timed", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    ("nmf_solver_update_h", C.c_int, [C.c_void_p]),
    ("nmf_solver_update_w", C.c_int, [C.c_void_p]),
    ("nmf_solver_check", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("nmf_solver_check_sums", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("nmf_solver_run", C.c_int, [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(_result)]),
    ("nmf_solver_sync", C.c_int, [C.c_void_p]),
    ("nmf_solver_time_piece", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("nmf_solver_w_partial", C.c_int, [C.c_void_p]),
    ("nmf_solver_w_apply", C.c_int, [C.c_void_p]),
    ("nmf_solver_partial_buffer", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("nmf_solver_set_partial_buffer", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("nmf_solver_create_batched", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_opts)]),
    ("nmf_solver_batch", C.c_int, [C.c_void_p]),
    ("nmf_solver_upload_pair", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("nmf_solver_download_pair", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("nmf_solver_check_pair", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("nmf_solver_set_active", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("nmf_solver_check_all", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("nmf_solver_uses_split_kernel", C.c_int, [C.c_void_p]),
    ("nmf_solver_path", C.c_int, [C.c_void_p]),
    ("nmf_solver_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("nmf_solver_stream", C.c_void_p, [C.c_void_p]),
    ("nmf_comm_get_unique_id", C.c_int, [C.c_char_p]),
    ("nmf_comm_init_rank", C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int]),
    ("nmf_comm_destroy", None, [C.c_void_p]),
    ("nmf_comm_probe", C.c_int, [C.c_void_p, C.c_double]),
    ("nmf_comm_library_info", C.c_int, [C.c_char_p, C.c_int]),
    ("nmf_comm_capture_refusals", C.c_long, []),
    ("nmf_worth_sharding", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("nmf_plan_describe", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_opts), C.c_char_p, C.c_int]),
    ("nmf_debug_record_kernels", C.c_int, [C.c_int]),
    ("nmf_debug_last_kernel", C.c_char_p, []),
    ("nmf_device_count", C.c_int, []),
    ("nmf_device_name", C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    ("nmf_version", C.c_char_p, []),
]


def declared_symbols():
    return [s[0] for s in _SIGS]


_lib = None
_loaded_before_torch = False


def lib():
    """Load libnmf_mi355x.so; raises if it has not been built (no fallback).

    PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If torch is imported first, this
    library's HIP symbols bind to that same runtime (one runtime per process: streams and pointers interoperate);
    if this library is loaded first and torch later, the process ends up with two HIP runtimes.  Code that shares
    streams with torch (GpuShard, bench.py for N > 1) therefore imports torch before the first call in here."""
    global _lib, _loaded_before_torch
    if _lib is None:
        import sys
        _loaded_before_torch = "torch" not in sys.modules
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C nmf-gpu_amd/csrc)")
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SIGS:
            f = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _chk(st: int):
    if st != 0:
        raise NmfError(st, f"{lib().nmf_status_string(st).decode()} ({lib().nmf_last_error().decode()})")


def device_count() -> int:
    return int(lib().nmf_device_count())


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    _chk(lib().nmf_device_name(device, buf, 256))
    return buf.value.decode()


# ------------------------------------------------------------------------------ Matrix
class Matrix:
    """Column-major fp32 matrix: host numpy buffer + optional device mirror
    (reference ``class Matrix``, cuda/matrix.cuh:18-39; ``matrix`` struct, README.md:31-36)."""

    def __init__(self, data=None, rows: Optional[int] = None, cols: Optional[int] = None, value: float = 0.0):
        if data is not None:
            a = np.asarray(data, dtype=np.float32)
            if a.ndim != 2:
                raise ValueError("Matrix needs a 2-D array")
            self.mat = np.asfortranarray(a).copy(order="F")
        else:
            if rows is None or cols is None or rows <= 0 or cols <= 0:
                raise ValueError("Matrix needs positive rows/cols")
            self.mat = np.full((rows, cols), value, dtype=np.float32, order="F")
        self._c = _matrix()
        self._c.mat = self.mat.ctypes.data_as(_f32p)
        self._c.mat_d = None
        self._c.dim[0], self._c.dim[1] = self.mat.shape

    rows = property(lambda self: int(self._c.dim[0]))
    cols = property(lambda self: int(self._c.dim[1]))
    shape = property(lambda self: (self.rows, self.cols))

    @property
    def on_device(self) -> bool:
        return bool(self._c.mat_d)

    def to_device(self) -> "Matrix":
        _chk(lib().nmf_matrix_to_device(C.byref(self._c)))
        return self

    def from_device(self) -> "Matrix":
        _chk(lib().nmf_matrix_from_device(C.byref(self._c)))
        return self

    def free_device(self) -> None:
        if self._c.mat_d:
            _chk(lib().nmf_matrix_free_device(C.byref(self._c)))

    def numpy(self) -> np.ndarray:
        return self.mat

    def __del__(self):
        try:
            if self._c.mat_d and _lib is not None:
                _lib.nmf_matrix_free_device(C.byref(self._c))
        except Exception:
            pass


def _as_matrix(a) -> Matrix:
    return a if isinstance(a, Matrix) else Matrix(a)


def _inout(a, who: str):
    """An in/out factor (W or H of update_div): a ``Matrix`` is updated through its own buffer; a numpy array is copied
    into a column-major ``Matrix`` for the call and the result is copied back into the caller's array by ``_copy_back``;
    anything else cannot be updated in place and is refused."""
    if isinstance(a, Matrix):
        return a, None
    if not isinstance(a, np.ndarray):
        raise TypeError(f"{who} is updated in place: pass an nmf Matrix or a numpy array, not {type(a).__name__}")
    if not a.flags.writeable:
        raise TypeError(f"{who} is updated in place: the array is read-only")
    return Matrix(a), a


def _copy_back(pairs):
    for m, orig in pairs:
        if orig is not None:
            np.copyto(orig, m.mat, casting="same_kind")


def read_matrix(path: str) -> Matrix:
    """cuda/nmf.cu:188-218 (header uint32 rows, cols; column-major float32 payload)."""
    m = _matrix()
    _chk(lib().nmf_read_matrix(C.byref(m), path.encode()))
    try:
        n = m.dim[0] * m.dim[1]
        arr = np.ctypeslib.as_array(m.mat, shape=(n,)).copy().reshape((m.dim[0], m.dim[1]), order="F")
    finally:
        lib().nmf_destroy_matrix(C.byref(m))
    return Matrix(arr)


def write_matrix(A, path: str) -> None:
    """cuda/nmf.cu:220-259."""
    A = _as_matrix(A)
    _chk(lib().nmf_write_matrix(A._c, path.encode()))


# ------------------------------------------------------------------------------ update_div
def _make_opts(**kw) -> _opts:
    o = _opts()
    lib().nmf_default_opts(C.byref(o))
    for k, v in kw.items():
        if v is None:
            continue
        if k == "devices":      # a Python list of HIP ordinals; the array must outlive the call: kept on the struct
            arr = (C.c_int * len(v))(*[int(d) for d in v])
            o._devices_keepalive = arr
            o.devices = C.cast(arr, C.POINTER(C.c_int))
            continue
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k}")
        setattr(o, k, v)
    return o


def _result_dict(r: _result) -> dict:
    return {"iterations": int(r.iterations), "kl": [float(r.kl[i]) for i in range(r.n_kl)],
            "rel_l1": float(r.rel_l1), "path_used": int(r.path_used), "n_shards": int(r.n_shards),
            "w_replicas_identical": int(r.w_replicas_identical),
            "t": {T_NAMES[i]: float(r.t[i]) for i in range(10)}}


def update_div(W, H, X, CONVERGE_THRESH: float = 0.0, max_iter: int = 200, t=None, verbose: int = 0) -> None:
    """The documented drop-in (README.md:40-54): W, H updated in place (their ``.mat``),
    ``t`` an optional list/array of >= 10 doubles that receives the timers."""
    (W, w0), (H, h0), X = _inout(W, "W"), _inout(H, "H"), _as_matrix(X)
    if H.rows != W.cols or X.rows != W.rows or X.cols != H.cols:
        # the C entry point exit()s on a shape error like the reference; raise here instead
        raise NmfError(2, "dimensions do not agree")
    tt = (C.c_double * 10)() if t is not None else None
    lib().update_div(W._c, H._c, X._c, CONVERGE_THRESH, max_iter, tt, verbose)
    _copy_back(((W, w0), (H, h0)))
    if t is not None:
        for i in range(10):
            t[i] = tt[i]


def update_div_ex(W, H, X, **opts) -> dict:
    """Status-returning variant; options as in ``nmf_opts``.  W, H (``Matrix`` or numpy arrays) are updated in place.
    Returns the result dict."""
    (W, w0), (H, h0), X = _inout(W, "W"), _inout(H, "H"), _as_matrix(X)
    o = _make_opts(**opts)
    r = _result()
    _chk(lib().update_div_ex(W._c, H._c, X._c, C.byref(o), C.byref(r)))
    _copy_back(((W, w0), (H, h0)))
    return _result_dict(r)


def update_div_restarts(Ws, Hs, X, **opts):
    """Paper section 3.2: run every (W, H) initialisation, keep X resident; returns (best_index, [kl...]).
    All pairs are updated in place."""
    wp = [_inout(w, "W") for w in Ws]
    hp = [_inout(h, "H") for h in Hs]
    Ws, Hs = [m for m, _ in wp], [m for m, _ in hp]
    X = _as_matrix(X)
    n = len(Ws)
    if n == 0 or len(Hs) != n:
        raise NmfError(1, "need the same positive number of W and H initialisations")
    wa = (_matrix * n)(*[w._c for w in Ws])
    ha = (_matrix * n)(*[h._c for h in Hs])
    o = _make_opts(**opts)
    best = C.c_int(-1)
    kl = (C.c_double * n)()
    _chk(lib().update_div_restarts(wa, ha, n, X._c, C.byref(o), C.byref(best), kl))
    _copy_back(wp + hp)
    return best.value, [kl[i] for i in range(n)]


# ------------------------------------------------------------------------------ operators
def _op3(name, a, b, c, stream=None):
    _chk(getattr(lib(), name)(a._c, b._c, c._c, stream))


def matrix_multiply(a, b, c, stream=None): _op3("nmf_matrix_multiply", a, b, c, stream)
def matrix_multiply_AtB(a, b, c, stream=None): _op3("nmf_matrix_multiply_AtB", a, b, c, stream)
def matrix_multiply_ABt(a, b, c, stream=None): _op3("nmf_matrix_multiply_ABt", a, b, c, stream)
def element_multiply(a, b, c, stream=None): _op3("nmf_element_multiply", a, b, c, stream)
def element_divide(a, b, c, stream=None): _op3("nmf_element_divide", a, b, c, stream)
def row_divide(a, b, c, stream=None): _op3("nmf_row_divide", a, b, c, stream)
def col_divide(a, b, c, stream=None): _op3("nmf_col_divide", a, b, c, stream)


def set_epsilon(a, stream=None):
    _chk(lib().nmf_set_epsilon(a._c, stream))


def sum_cols(a, out, stream=None):
    _chk(lib().nmf_sum_cols(a._c, out._c, stream))


def sum_rows(a, out, stream=None):
    _chk(lib().nmf_sum_rows(a._c, out._c, stream))


def kl_divergence(x, y, stream=None) -> float:
    v = C.c_double()
    _chk(lib().nmf_kl_divergence(x._c, y._c, C.byref(v), stream))
    return v.value


def diff_norm(x, y, stream=None):
    d, a = C.c_double(), C.c_double()
    _chk(lib().nmf_diff_norm(x._c, y._c, C.byref(d), C.byref(a), stream))
    return d.value, a.value


# ------------------------------------------------------------------------------ Comm / Solver
class Comm:
    """RCCL communicator handle for N-sharded runs (one process per GPU)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _chk(lib().nmf_comm_get_unique_id(buf))
        return buf.raw

    def __init__(self, uid: bytes, rank: int, nranks: int):
        self._h = C.c_void_p()
        _chk(lib().nmf_comm_init_rank(C.byref(self._h), uid, rank, nranks))
        self.rank, self.nranks = rank, nranks

    def probe(self, timeout_s: float = 0.0) -> None:
        """collective: one small all-reduce with a deadline, checked; raises NmfError if the communicator does not work"""
        _chk(lib().nmf_comm_probe(self._h, float(timeout_s)))

    def close(self):
        if self._h:
            lib().nmf_comm_destroy(self._h)
            self._h = C.c_void_p()


def comm_library_info() -> str:
    """version and path of the RCCL actually loaded (loads it), and the rccl.h version the library was built against"""
    buf = C.create_string_buffer(768)
    lib().nmf_comm_library_info(buf, 768)
    return buf.value.decode()


def comm_capture_refusals() -> int:
    """process-wide count of waits whose closing hipStreamSynchronize was refused because of a capture elsewhere (expected: 0)"""
    return int(lib().nmf_comm_capture_refusals())


def plan_describe(M: int, N: int, K: int, batch: int = 1, **opts) -> str:
    """the describe() line of the solver that Solver(M, N, K, batch=batch, **opts) would create; touches no device"""
    o = _make_opts(**opts)
    buf = C.create_string_buffer(256)
    _chk(lib().nmf_plan_describe(M, N, K, batch, C.byref(o), buf, 256))
    return buf.value.decode()


def record_kernels(on: bool = True) -> bool:
    """switch the launchers' kernel-name recording on or off; returns the previous setting"""
    return bool(lib().nmf_debug_record_kernels(int(on)))


def last_kernel() -> str:
    """demangled name of the fused half-step / check kernel this thread launched last (with recording on)"""
    return lib().nmf_debug_last_kernel().decode()


def _hostptr(a, rows, cols):
    if a is None:
        return None, None
    arr = np.asfortranarray(np.asarray(a, dtype=np.float32))
    if arr.shape != (rows, cols):
        raise NmfError(2, f"dimensions do not agree: got {arr.shape}, want {(rows, cols)}")
    return arr, arr.ctypes.data_as(C.c_void_p)


class Solver:
    """W, H, X resident in HBM; ``iterate`` enqueues update_h + update_w pairs
    (``run_async``, cuda/nmf.cu:76-116)."""

    def __init__(self, M: int, N: int, K: int, *, path: int = PATH_AUTO, use_graph: bool = True,
                 device: int = -1, stream: Optional[int] = None, comm: Optional[Comm] = None,
                 nsplit_h: int = 0, nsplit_w: int = 0, fast_divide: int = 0, split_kernel: int = 0, batch: int = 1):
        self.M, self.N, self.K, self.batch = M, N, K, batch
        o = _make_opts(path=path, use_graph=int(use_graph), device=device, stream=stream,
                       comm=(comm._h.value if comm is not None else None), nsplit_h=nsplit_h, nsplit_w=nsplit_w,
                       fast_divide=int(fast_divide), split_kernel=int(split_kernel))
        self._h = C.c_void_p()
        self._comm = comm
        _chk(lib().nmf_solver_create_batched(C.byref(self._h), M, N, K, batch, C.byref(o)))

    def close(self):
        if self._h:
            lib().nmf_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def path(self) -> int:
        return int(lib().nmf_solver_path(self._h))

    @property
    def stream(self) -> int:
        return int(lib().nmf_solver_stream(self._h) or 0)

    def upload(self, W=None, H=None, X=None):
        keep = []
        ptrs = []
        for a, (r, c) in ((W, (self.M, self.K)), (H, (self.K, self.N)), (X, (self.M, self.N))):
            arr, p = _hostptr(a, r, c)
            keep.append(arr)
            ptrs.append(p)
        _chk(lib().nmf_solver_upload(self._h, *ptrs))

    def upload_device(self, W_ptr=None, H_ptr=None, X_ptr=None):
        """Unpadded column-major device buffers (e.g. ``tensor.data_ptr()``)."""
        _chk(lib().nmf_solver_upload_device(self._h, W_ptr, H_ptr, X_ptr))

    def download(self):
        W = np.empty((self.M, self.K), dtype=np.float32, order="F")
        H = np.empty((self.K, self.N), dtype=np.float32, order="F")
        _chk(lib().nmf_solver_download(self._h, W.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)))
        return W, H

    def describe(self) -> str:
        buf = C.create_string_buffer(256)
        _chk(lib().nmf_solver_describe(self._h, buf, 256))
        return buf.value.decode()

    @property
    def uses_split_kernel(self) -> bool:
        return bool(lib().nmf_solver_uses_split_kernel(self._h))

    # pair b of a batched solver (multi-restart: B (W, H) pairs against one X per launch)
    def upload_pair(self, b: int, W=None, H=None):
        wa, wp = _hostptr(W, self.M, self.K)
        ha, hp = _hostptr(H, self.K, self.N)
        _chk(lib().nmf_solver_upload_pair(self._h, b, wp, hp))

    def download_pair(self, b: int):
        W = np.empty((self.M, self.K), dtype=np.float32, order="F")
        H = np.empty((self.K, self.N), dtype=np.float32, order="F")
        _chk(lib().nmf_solver_download_pair(self._h, b, W.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)))
        return W, H

    def check_pair(self, b: int):
        kl, rl1 = C.c_double(), C.c_double()
        _chk(lib().nmf_solver_check_pair(self._h, b, C.byref(kl), C.byref(rl1)))
        return kl.value, rl1.value

    def check_all(self):
        kl, rl1 = (C.c_double * self.batch)(), (C.c_double * self.batch)()
        _chk(lib().nmf_solver_check_all(self._h, kl, rl1))
        return list(kl), list(rl1)

    def set_active(self, flags=None):
        arr = None if flags is None else (C.c_int * self.batch)(*[int(bool(f)) for f in flags])
        _chk(lib().nmf_solver_set_active(self._h, arr))

    def iterate(self, iters: int = 1):
        _chk(lib().nmf_solver_iterate(self._h, iters))

    def prepare(self, iters: int):
        """capture the hipGraphs an iterate(iters) call replays, without running them"""
        _chk(lib().nmf_solver_prepare(self._h, iters))

    def iterate_timed(self, iters: int = 1) -> dict:
        """eager iterations with hipEvents around every piece; returns device seconds per piece name"""
        t = (C.c_double * 10)()
        _chk(lib().nmf_solver_iterate_timed(self._h, iters, t))
        return {T_NAMES[i]: float(t[i]) for i in range(10)}

    def update_h(self):
        _chk(lib().nmf_solver_update_h(self._h))

    def update_w(self):
        _chk(lib().nmf_solver_update_w(self._h))

    def w_partial(self):
        _chk(lib().nmf_solver_w_partial(self._h))

    def w_apply(self):
        _chk(lib().nmf_solver_w_apply(self._h))

    def partial_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(lib().nmf_solver_partial_buffer(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def set_partial_buffer(self, dev_ptr: int, count: int):
        _chk(lib().nmf_solver_set_partial_buffer(self._h, dev_ptr, count))

    def check(self):
        kl, rl1 = C.c_double(), C.c_double()
        _chk(lib().nmf_solver_check(self._h, C.byref(kl), C.byref(rl1)))
        return kl.value, rl1.value

    def check_sums(self):
        v = (C.c_double * 3)()
        _chk(lib().nmf_solver_check_sums(self._h, v))
        return v[0], v[1], v[2]

    def run(self, thresh: float = 0.0, max_iter: int = 200, iter_check: int = 25, verbose: int = 0) -> dict:
        r = _result()
        _chk(lib().nmf_solver_run(self._h, thresh, max_iter, iter_check, verbose, C.byref(r)))
        return _result_dict(r)

    def sync(self):
        _chk(lib().nmf_solver_sync(self._h))

    def time_piece(self, which: int, reps: int = 5) -> float:
        ms = C.c_double()
        _chk(lib().nmf_solver_time_piece(self._h, which, reps, C.byref(ms)))
        return ms.value
