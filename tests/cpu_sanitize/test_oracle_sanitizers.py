"""The oracle's C code under AddressSanitizer + UBSan, on the CPU.  Lives under tests/cpu_sanitize/, which .gpurunignore keeps off the
GPU box: the pool runs no sanitizer builds, and refuses a push that carries a file asking for one (tests/test_gpu_push_guard.py)."""
import os
import subprocess
import sys


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """SURVEY 5 (race / memory checking): the GPU pool has no sanitizer support, so the CPU side is what can be checked -- the
    oracle's C code (the checker every parity claim rests on) built with -fsanitize=address,undefined and driven through its
    loop, half-steps, GEMMs, sums, generator and file I/O on ragged shapes in a child process (ASan has to be loaded first)."""
    from conftest import ROOT
    so = tmp_path / "libnmf_oracle_san.so"
    src = [os.path.join(ROOT, "oracle", f) for f in ("nmf_oracle.c", "nmf_oracle_fast.c")]
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-fopenmp", "-mavx2", "-mfma", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-shared", "-o", str(so)] + src + ["-lm"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    code = f"""
import sys, os
sys.path.insert(0, {ROOT!r})
import numpy as np, oracle
oracle._lib = oracle._bind({str(so)!r})        # route every call through the sanitized build
for (M, N, K) in ((33, 47, 5), (64, 96, 32), (100, 70, 17), (1, 9, 1)):
    X, W, H = oracle.gen_problem(M, N, K, seed=3)
    w, h, it, kl = oracle.update_div(W, H, X, 1e-9, 7, 3)
    assert np.isfinite(w).all() and np.isfinite(h).all() and it >= 3
    w2, h2, _, _ = oracle.update_div(W, H, X, 0.0, 4, 25, oracle.MODE_REFCOMPAT)
    wf, hf = oracle.update_div_fast(W, H, X, 3)
    oracle.update_h(oracle.clamp(W), oracle.clamp(H), oracle.clamp(X)); oracle.update_w(oracle.clamp(W), oracle.clamp(H), oracle.clamp(X))
    for kind, A, B in (("nn", W, H), ("tn", W, X), ("nt", X, H)):
        oracle.sgemm(kind, A, B)
    oracle.sum_cols(W); oracle.sum_rows(H); oracle.kl_div(oracle.clamp(X), np.maximum(oracle.sgemm("nn", W, H), oracle.EPS))
    p = os.path.join({str(tmp_path)!r}, "a.bin"); oracle.write_bin(p, W); assert np.array_equal(oracle.read_bin(p), W)
print("sanitized oracle ok")
"""
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
