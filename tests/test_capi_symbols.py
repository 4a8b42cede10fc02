"""The C-ABI library loads without a GPU and exports every symbol include/nmf_mi355x.h declares;
host-only entry points behave like the reference's (no compute calls here)."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nmf_mi355x.h")


def _declared_in_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}()]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_symbols_all_exported_and_bound(ng):
    declared = _declared_in_header()
    assert "update_div" in declared and "nmf_solver_create" in declared and len(declared) > 40
    L = ng.lib()
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    # the ctypes mirror binds exactly the header's list
    assert sorted(ng.declared_symbols()) == declared
    out = subprocess.run(["nm", "-D", "--defined-only", ng.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    assert set(declared) <= exported


def test_no_oracle_or_torch_dependency_in_product(ng):
    out = subprocess.run(["ldd", ng.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "oracle" not in out and "torch" not in out and "libamdhip64" in out
    pkg = os.path.join(ROOT, "nmf-gpu_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "import oracle" not in txt and "nmf_oracle" not in txt and "from oracle" not in txt, fn


def test_matrix_struct_layout(ng):
    """matrix = {float* mat; float* mat_d; int dim[2];} (README.md:31-36)"""
    from nmf_gpu_amd.api import _matrix
    assert C.sizeof(_matrix) == 24 and _matrix.mat.offset == 0 and _matrix.mat_d.offset == 8 and _matrix.dim.offset == 16


def test_opts_and_result_struct_layouts_match_the_header(ng, tmp_path):
    """the ctypes mirrors of nmf_opts / nmf_result against what a C compiler makes of include/nmf_mi355x.h"""
    from nmf_gpu_amd.api import _opts, _result
    src = tmp_path / "layout.c"
    fields_o = [f[0] for f in _opts._fields_]
    fields_r = [f[0] for f in _result._fields_]
    body = "".join(f'printf("o {f} %zu\\n", offsetof(nmf_opts, {f}));' for f in fields_o)
    body += "".join(f'printf("r {f} %zu\\n", offsetof(nmf_result, {f}));' for f in fields_r)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nmf_mi355x.h"\nint main(void){'
                   'printf("so %zu\\nsr %zu\\n", sizeof(nmf_opts), sizeof(nmf_result));' + body + "return 0;}")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)   # plain C: no HIP, no torch
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.startswith(("o ", "r "))}
    sizes = {l.split()[0]: int(l.split()[1]) for l in out if l.startswith(("so ", "sr "))}
    assert sizes == {"so": C.sizeof(_opts), "sr": C.sizeof(_result)}
    for f in fields_o:
        assert got[("o", f)] == getattr(_opts, f).offset, f
    for f in fields_r:
        assert got[("r", f)] == getattr(_result, f).offset, f


def test_read_write_matrix_format(ng, tmp_path):
    """uint32 rows, uint32 cols, float32 column-major, no padding (cuda/nmf.cu:194-204, 239-249)"""
    A = np.asfortranarray(np.arange(15, dtype=np.float32).reshape(3, 5))
    p = str(tmp_path / "m.bin")
    ng.write_matrix(ng.Matrix(A), p)
    raw = open(p, "rb").read()
    assert struct.unpack("II", raw[:8]) == (3, 5)
    assert np.array_equal(np.frombuffer(raw[8:], np.float32), A.reshape(-1, order="F"))
    B = ng.read_matrix(p)
    assert B.shape == (3, 5) and np.array_equal(B.mat, A)
    with pytest.raises(ng.NmfError) as e:
        ng.read_matrix(str(tmp_path / "missing.bin"))     # the reference never checks fopen (cuda/nmf.cu:196)
    assert e.value.status == 4
    open(str(tmp_path / "short.bin"), "wb").write(raw[:20])
    with pytest.raises(ng.NmfError):
        ng.read_matrix(str(tmp_path / "short.bin"))


def test_reads_reference_golden_files(ng):
    W = ng.read_matrix(os.path.join(ROOT, "tests", "golden", "Wtest.bin"))
    H = ng.read_matrix(os.path.join(ROOT, "tests", "golden", "Htest.bin"))
    assert W.shape == (4096, 128) and H.shape == (128, 350)
    assert abs(float(W.mat.max()) - 188.644) < 1e-2 and int((H.mat == 0).sum()) == 38850


def test_shape_errors_before_any_gpu_work(ng):
    """dimension mismatch -> NMF_ERR_SHAPE (the reference prints and exit(1)s, cuda/matrix.cu:130-134)"""
    with pytest.raises(ng.NmfError) as e:
        ng.update_div_ex(ng.Matrix(rows=4, cols=2), ng.Matrix(rows=3, cols=5), ng.Matrix(rows=4, cols=5))
    assert e.value.status == 2
    with pytest.raises(ng.NmfError) as e:
        ng.update_div(ng.Matrix(rows=4, cols=2), ng.Matrix(rows=2, cols=5), ng.Matrix(rows=4, cols=6), 0.0, 1)
    assert e.value.status == 2


def test_empty_matrices_are_rejected(ng):
    """zero-sized / NULL inputs -> NMF_ERR_ARG, never a launch (the reference would cudaMalloc(0) and launch empty grids)"""
    import ctypes as C
    from nmf_gpu_amd.api import _matrix, _opts, _result
    buf = (C.c_float * 4)()
    def m(rows, cols, data=True):
        x = _matrix(); x.mat = C.cast(buf, C.POINTER(C.c_float)) if data else None; x.mat_d = None
        x.dim[0], x.dim[1] = rows, cols
        return x
    L = ng.lib()
    assert L.update_div_ex(m(0, 2), m(2, 2), m(0, 2), None, None) == 1
    assert L.update_div_ex(m(2, 2), m(2, 0), m(2, 0), None, None) == 1
    assert L.update_div_ex(m(2, 2, data=False), m(2, 2), m(2, 2), None, None) == 1
    h = C.c_void_p()
    assert L.nmf_solver_create(C.byref(h), 0, 4, 4, None) == 1 and L.nmf_solver_create(C.byref(h), 4, 4, -1, None) == 1
    with pytest.raises(ValueError):
        ng.Matrix(rows=0, cols=3)


def test_fails_loudly_without_gpu_or_library(ng, monkeypatch):
    if ng.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ng.NmfError) as e:     # no silent CPU fallback
        ng.Solver(64, 64, 8)
    assert e.value.status == 3
    import nmf_gpu_amd.api as api
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "LIB_PATH", "/nonexistent/libnmf_mi355x.so")
    with pytest.raises(ImportError):
        api.lib()


def test_defaults_follow_reference_knobs(ng):
    from nmf_gpu_amd.api import _opts
    o = _opts()
    ng.lib().nmf_default_opts(C.byref(o))
    assert o.max_iter == 200 and o.iter_check == 25 and o.converge_thresh == 0.0   # cuda/nmf.cu:9-11
    assert o.use_graph == 2 and o.path == ng.PATH_AUTO      # NMF_GRAPH_AUTO
    assert ng.lib().nmf_status_string(2) == b"dimensions do not agree"


def test_cli_built_and_usage(ng):
    cli = os.path.join(ROOT, "nmf-gpu_amd", "nmf")
    assert os.path.exists(cli)
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--iters" in r.stderr
