"""The CLI's CPU-only tools: `nmf generate` restates matrix_export.py byte for byte, `nmf compare` is
test_output.sh with a tolerance.  Neither touches the GPU."""
import hashlib
import os
import subprocess

import numpy as np

from conftest import GOLDEN, ROOT
from test_oracle_golden import REF_MD5

CLI = os.path.join(ROOT, "nmf-gpu_amd", "nmf")


def _run(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120, cwd=cwd)


def test_generate_reproduces_the_reference_inputs(tmp_path):
    """default shape and seed = matrix_export.py:4-7: the md5s of the files that script writes"""
    r = _run("generate", cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    for name, md5 in REF_MD5.items():
        assert hashlib.md5((tmp_path / f"{name}.bin").read_bytes()).hexdigest() == md5


def test_generate_other_shapes_match_the_oracle_generator(oracle, tmp_path):
    M, N, K, seed = 37, 53, 11, 5
    r = _run("generate", "--M", str(M), "--N", str(N), "--K", str(K), "--seed", str(seed),
             "--X", str(tmp_path / "x.bin"), "--W", str(tmp_path / "w.bin"), "--H", str(tmp_path / "h.bin"))
    assert r.returncode == 0, r.stderr
    X, W, H = oracle.gen_problem(M, N, K, seed=seed)
    for f, A in (("x.bin", X), ("w.bin", W), ("h.bin", H)):
        assert np.array_equal(oracle.read_bin(str(tmp_path / f)), A)
    assert _run("generate", "--M", "0").returncode != 0


def test_compare_is_test_output_with_a_tolerance(oracle, tmp_path):
    A = np.asfortranarray(np.random.default_rng(0).random((40, 30), dtype=np.float32))
    B = A * np.float32(1 + 3e-5)
    oracle.write_bin(str(tmp_path / "a.bin"), A); oracle.write_bin(str(tmp_path / "b.bin"), B)
    r = _run("compare", str(tmp_path / "a.bin"), str(tmp_path / "a.bin"))
    assert r.returncode == 0 and "bytes identical" in r.stdout
    r = _run("compare", str(tmp_path / "b.bin"), str(tmp_path / "a.bin"))
    assert r.returncode == 0 and "bytes differ" in r.stdout and "matches" in r.stdout          # 3e-5 <= 1e-4
    r = _run("compare", str(tmp_path / "b.bin"), str(tmp_path / "a.bin"), "--tol", "1e-6")
    assert r.returncode == 1 and "differs" in r.stdout
    oracle.write_bin(str(tmp_path / "c.bin"), np.asfortranarray(A[:, :7]))
    assert _run("compare", str(tmp_path / "c.bin"), str(tmp_path / "a.bin")).returncode == 1  # shapes differ
    r = _run("compare", str(tmp_path / "nope.bin"), str(tmp_path / "a.bin"))
    assert r.returncode != 0 and "cannot open" in r.stderr
    # the reference's golden pair against itself, and against each other's inputs
    assert _run("compare", os.path.join(GOLDEN, "Wtest.bin"), os.path.join(GOLDEN, "Wtest.bin")).returncode == 0
