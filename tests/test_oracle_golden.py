"""Pins the CPU oracle (test infrastructure) against everything the reference's own tests hold
for this path: the input generator (matrix_export.py:4-12, md5s recorded in SURVEY 4.1 and
re-derived by running the script in this container) and the golden outputs Wtest.bin / Htest.bin
(test_output.sh:5-18).  No GPU involved."""
import hashlib
import os
import struct

import numpy as np

from conftest import GOLDEN

# md5 of X.bin / W.bin / H.bin as written by the reference's matrix_export.py (numpy 2.2.6)
REF_MD5 = {"X": "1ab4ccb0efd4c4a9db9d9acce696ee11", "W": "68afe6b2626985e55432776061046728",
           "H": "516a8880ad486a54dc1049757087f038"}
GOLD_MD5 = {"Wtest.bin": "76f9988f", "Htest.bin": "0f6eaee4"}   # prefixes quoted in SURVEY 2 (#8)


def _bin_md5(A):
    return hashlib.md5(struct.pack("ii", *A.shape) + A.reshape(-1, order="F").tobytes()).hexdigest()


def test_generator_reproduces_reference_inputs(oracle):
    X, W, H = oracle.gen_problem(4096, 350, 128, seed=0)
    assert _bin_md5(X) == REF_MD5["X"] and _bin_md5(W) == REF_MD5["W"] and _bin_md5(H) == REF_MD5["H"]
    flat = X.reshape(-1, order="F")
    assert np.allclose(flat[:4], [0.548813522, 0.715189338, 0.602763355, 0.544883192], rtol=0, atol=1e-9)
    assert abs(float(X.sum(dtype=np.float64)) - 717076.779294) < 1e-3
    # and it is the same stream numpy's legacy RandomState produces
    rs = np.random.RandomState(0)
    assert np.array_equal(flat[:1000], rs.rand(1000).astype(np.float32))


def test_golden_fixtures_intact():
    for name, pre in GOLD_MD5.items():
        with open(os.path.join(GOLDEN, name), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest().startswith(pre)


def test_refcompat_reproduces_reference_golden_outputs(oracle):
    """refcompat = the intended loop minus row_divide (invalid launch, cuda/matrix.cu:215-217) with
    sum_cols returning 14 of 128 partials (cuda/matrix.cu:676-683), 200 iterations.

    Tolerance, plainly: BASELINE.md's gate for this configuration is 1e-4 on both factors, and it is NOT met on H -- the oracle
    lands 6.8e-5 (W) and 4.3e-4 (H) from the reference's files, asserted below at 1e-4 / 6e-4.  Why it cannot be met: the gold run
    is a winner-take-all dynamic (17 of 128 components survive) that amplifies fp32 summation-order noise, and the reference's
    sums ran in closed-source cuBLAS whose order is unknowable; the same model evaluated in fp32 with the K components merely
    relabelled lands 8e-5..1.8e-4 (H) from the unpermuted run and 2.7e-4..4.3e-4 from the gold (next test: measured, not allowed).
    The discrete structure must match exactly: zero counts and the set of surviving components."""
    X, W, H = oracle.gen_problem(4096, 350, 128, seed=0)
    w, h, it, _ = oracle.update_div(W, H, X, 0.0, 200, 25, oracle.MODE_REFCOMPAT)
    Wg = oracle.read_bin(os.path.join(GOLDEN, "Wtest.bin"))
    Hg = oracle.read_bin(os.path.join(GOLDEN, "Htest.bin"))
    assert it == 200 and Wg.shape == (4096, 128) and Hg.shape == (128, 350)
    eW, eH = oracle.relF(w, Wg), oracle.relF(h, Hg)
    print("refcompat vs gold: relF(W)=%.3g relF(H)=%.3g" % (eW, eH))
    assert eW < 1e-4 and eH < 6e-4       # measured 6.8e-5 / 4.3e-4: H misses BASELINE.md's 1e-4; why: test_gold_distance_is_summation_order_noise
    assert abs(int((w == 0).sum()) - 457204) <= 5 and int((h == 0).sum()) == 38850
    surv = np.flatnonzero(h.sum(axis=1) > 0)
    assert list(surv) == [1, 22, 30, 34, 43, 45, 57, 70, 88, 89, 93, 96, 110, 113, 118, 121, 126]
    assert list(np.flatnonzero(Hg.sum(axis=1) > 0)) == list(surv)
    # sum(W) is conserved = sum(X) when row_divide never runs
    assert abs(float(w.sum(dtype=np.float64)) - float(np.maximum(X, oracle.EPS).sum(dtype=np.float64))) < 1.0


def test_gold_distance_is_summation_order_noise(oracle):
    """The reference's four GEMMs ran in closed-source cuBLAS (cuda/matrix.cu:101-124), whose summation order over k is
    unknowable, so bit parity with Wtest/Htest is impossible by construction.  This test turns the tolerance above from an
    allowance into a measurement: the SAME model (refcompat) is run with the K components relabelled -- columns of W and
    rows of H permuted, results un-permuted -- which changes nothing mathematically, only the order in which fp32 sums over
    k are taken.  After 200 iterations of the winner-take-all dynamic two such runs differ from each other by 2.5e-5..4.9e-5
    (W) and 8e-5..1.8e-4 (H), and each is 6.8e-5..9.9e-5 (W) / 2.7e-4..4.3e-4 (H) from the gold (measured here, 8 threads).
    Asserted: (i) order alone moves the result by more than 1e-5; (ii) the gold lies within 4x the largest distance between
    two order-perturbed runs of our own model, i.e. it is one more member of that cloud, not a different model; (iii) the
    discrete outcome (which entries of H are exactly zero, which 17 components survive) is identical in every run."""
    X, W, H = oracle.gen_problem(4096, 350, 128, seed=0)
    Wg = oracle.read_bin(os.path.join(GOLDEN, "Wtest.bin"))
    Hg = oracle.read_bin(os.path.join(GOLDEN, "Htest.bin"))
    perms = [np.arange(128), np.arange(128)[::-1].copy(), np.random.default_rng(1).permutation(128)]
    runs = []
    for p in perms:
        w, h, it, _ = oracle.update_div(np.asfortranarray(W[:, p]), np.asfortranarray(H[p, :]), X, 0.0, 200, 25, oracle.MODE_REFCOMPAT)
        inv = np.argsort(p)
        runs.append((np.asfortranarray(w[:, inv]), np.asfortranarray(h[inv, :])))
    gold = [(oracle.relF(w, Wg), oracle.relF(h, Hg)) for w, h in runs]
    pair = [(oracle.relF(runs[i][0], runs[j][0]), oracle.relF(runs[i][1], runs[j][1])) for i in range(3) for j in range(i + 1, 3)]
    print("to gold:", gold, "pairwise:", pair)
    sW, sH = max(p[0] for p in pair), max(p[1] for p in pair)
    assert min(p[0] for p in pair) > 1e-5 and min(p[1] for p in pair) > 1e-5
    for eW, eH in gold:
        assert eW < 4 * sW and eH < 4 * sH and eW < 1.5e-4 and eH < 6e-4
    for w, h in runs:
        assert np.array_equal(h == 0, Hg == 0)
        assert list(np.flatnonzero(h.sum(axis=1) > 0)) == [1, 22, 30, 34, 43, 45, 57, 70, 88, 89, 93, 96, 110, 113, 118, 121, 126]


def test_spec_mode_is_not_the_golden_model(oracle):
    """The two bug hypotheses are what ties the oracle to the gold: the intended math is O(1) away."""
    X, W, H = oracle.gen_problem(4096, 350, 128, seed=0)
    w, h, _, _ = oracle.update_div(W, H, X, 0.0, 30, 25, oracle.MODE_SPEC)
    Wg = oracle.read_bin(os.path.join(GOLDEN, "Wtest.bin"))
    assert oracle.relF(w, Wg) > 0.5


def test_spec_kl_trajectory_known_answers(oracle):
    """SURVEY 4.1 KATs for the intended math on the gold inputs (fp32 numpy at survey time)."""
    X, W, H = oracle.gen_problem(4096, 350, 128, seed=0)
    _, h, it, kl = oracle.update_div(W, H, X, 0.0, 100, 25, oracle.MODE_SPEC)
    assert it == 100 and len(kl) == 5
    for got, want in zip(kl, [4.236641e7, 1.318188e5, 1.213339e5, None, 1.059821e5]):
        if want is not None:
            assert abs(got - want) / want < 1e-5, (got, want)
    assert all(kl[i + 1] < kl[i] for i in range(4))
    assert abs(float(h.sum(dtype=np.float64)) - 350.1) < 0.5   # sum(H) ~ 350.1 from iteration 1 on
