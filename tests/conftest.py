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


class _Cfg3OracleRun:
    """The CPU side of the cfg3 x 200 comparison -- 200 iterations of the oracle's fast arrangement, about two minutes on the box's 16 cores --
    run on a background thread (the C oracle releases the GIL) while the other tests of the module go on, so that the comparison itself only
    waits for what is not finished yet.  The same four calls of 50 iterations each, in the same order, as the test used to make itself; a
    cancelled session waits for the call in flight (up to ~30 s)."""

    def __init__(self, oracle, problem):
        import threading
        self._oracle, self._problem = oracle, problem
        self._cv = threading.Condition()
        self._stages, self._err, self._cancel, self.cpu_s = [], None, False, 0.0
        self._thread = threading.Thread(target=self._work, name="cfg3-oracle", daemon=True)
        self._thread.start()

    def _work(self):
        import time
        X, Wr, Hr = self._problem
        try:
            for _ in range(4):
                if self._cancel:
                    return
                t0 = time.time()
                Wr, Hr = self._oracle.update_div_fast(Wr, Hr, X, 50)
                self.cpu_s += time.time() - t0
                with self._cv:
                    self._stages.append((Wr, Hr))
                    self._cv.notify_all()
        except BaseException as e:      # handed to the waiting test
            with self._cv:
                self._err = e
                self._cv.notify_all()

    def stage(self, i):
        """(W, H) of the oracle after 50 (i + 1) iterations; blocks until the thread has got there"""
        with self._cv:
            while len(self._stages) <= i and self._err is None:
                self._cv.wait(timeout=1.0)
                if not self._thread.is_alive() and len(self._stages) <= i and self._err is None:
                    raise RuntimeError("the cfg3 oracle thread ended early")
            if self._err is not None:
                raise self._err
            return self._stages[i]

    def stop(self):
        self._cancel = True
        self._thread.join()


@pytest.fixture(scope="session")
def cfg3_oracle_200(oracle, cfg3_problem):
    run = _Cfg3OracleRun(oracle, cfg3_problem)
    yield run
    run.stop()


GOLDEN = os.path.join(ROOT, "tests", "golden")
