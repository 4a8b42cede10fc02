import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# update_div shards a large problem over every visible GPU by itself (nmf_opts.n_devices = 0).  The parity tests are about one
# GPU: on a multi-GPU host they must not turn into multi-GPU runs behind the test's back.  Tests of the multi-device driver ask
# for their ranks explicitly (n_devices / emulate_shards), which this does not affect.
os.environ.setdefault("NMF_DEVICES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built artefacts are git-ignored: a fresh checkout builds them (hipcc cross-compiles gfx950 without a GPU)
    pkg = os.path.join(ROOT, "nmf-gpu_amd")
    if not (os.path.exists(os.path.join(pkg, "libnmf_mi355x.so")) and os.path.exists(os.path.join(pkg, "nmf"))):
        import subprocess
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(pkg, "csrc"), "all"], check=True)


def _has_gpu() -> bool:
    try:
        import nmf_gpu_amd as ng
        return ng.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: only auto-skip when the
    # user did not ask for gpu tests explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def ng():
    import nmf_gpu_amd as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def cfg3_problem(oracle):
    """BASELINE config 3's seed-0 inputs (4096 x 65536 x 256: the generator needs ~8 s for the 268 M entries of X), drawn once
    per session and shared by the cfg3 tests.  Read-only: a test that changes them must copy."""
    X, W, H = oracle.gen_problem(4096, 65536, 256, seed=0)
    for a in (X, W, H):
        a.setflags(write=False)
    return X, W, H


GOLDEN = os.path.join(ROOT, "tests", "golden")
