"""Every kernel instantiation the half-step and check dispatchers can pick (launch_fused16, launch_split_step,
launch_fused_pair, the 32-column kernel for K <= 32) is launched BY NAME and compared with the oracle's half-step
(oracle/nmf_oracle.c: update_h / update_w following cuda/nmf.cu:118-176).  VERDICT r02 weak 4: the inline-asm MFMA chains are
correct only while their operands sit in VGPRs with the right wait states around them -- a property of the generated code of
EACH instantiation -- so each one is executed against the oracle here (tools/asm_audit.py inspects the same code statically).
The library reports the instantiation it launched (nmf_debug_last_kernel: hipKernelNameRefByPtr of the launched pointer)."""
import itertools
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 5e-6


def _args(name):
    m = re.search(r"nmf::(\w+)<([^>]*)>", name)
    assert m, name
    return m.group(1), tuple(a.strip() for a in m.group(2).split(","))


def _half_steps(ng, oracle, M, N, K, seen, **kw):
    """one H half-step, one W half-step and one check on a fresh problem; returns nothing, asserts parity, notes the kernels"""
    X, W, H = oracle.gen_problem(M, N, K, seed=K + M)
    s = ng.Solver(M, N, K, use_graph=False, **kw)
    s.upload(W, H, X)
    s.update_h()
    seen.add(_args(ng.last_kernel()))
    W1, H1 = s.download()
    Wc, Hc, Xc = oracle.clamp(W), oracle.clamp(H), oracle.clamp(X)
    Hr = oracle.update_h(Wc, Hc, Xc)
    eh = oracle.relF(H1, Hr)
    s.update_w()
    seen.add(_args(ng.last_kernel()))
    W2, _ = s.download()
    Wr = oracle.update_w(Wc, Hr, Xc)
    ew = oracle.relF(W2, Wr)
    kl, _ = s.check()
    seen.add(_args(ng.last_kernel()))
    klr = oracle.kl_div(Xc, oracle.clamp(oracle.sgemm("nn", Wr, Hr)))
    s.close()
    assert eh < TOL and ew < TOL, (ng.last_kernel(), kw, eh, ew)
    assert abs(kl - klr) <= 2e-5 * abs(klr), (kl, klr)


@pytest.fixture()
def recording(ng):
    old = ng.record_kernels(True)
    yield
    ng.record_kernels(old)


K16_KTS = tuple(range(1, 37))   # every multiple of 16 from K = 16 to 576 (round 5: KT = 33 .. 36, so that K just above 512 stays on this kernel)


def test_every_instantiation_of_the_64_column_kernel(ng, oracle, recording):
    """fused_step_kernel_k16<KT, WSTEP, PARTIAL, DIV, CHECK, OCC, GEMM, TRIM>: KT = K/16 in 1..16 (OCC = 2) and 17..36 (OCC = 1),
    both half-steps, in-place and partial-slab epilogues, and the CHECK instantiation of every KT (DIV = 0 throughout: the DIV = 1
    instantiations were removed in round 5, nmf_opts.fast_divide = 1 launches the same kernels -- asserted for one KT at the end).  The odd KT
    (K = 48, 80, ... 240: a remainder block in the k map, a zero-padded half piece in the LDS image) are the round-4 additions, as
    are the TRIM = 2, 3 variants (K <= 256, K % 64 != 0): a caller's K that leaves the last two / three steps of product 1 on zero
    padding (K = 16 KT - 8 / - 12 here: K = 100 on the K = 112 kernel) launches the chain that ends that many steps early (K = 16
    has four steps in all: its chain ends two early for K <= 8, never three)."""
    seen = set()
    for kt, ns in itertools.product(K16_KTS, (1, 2)):
        _half_steps(ng, oracle, 160, 208, 16 * kt, seen, split_kernel=-1, nsplit_h=ns, nsplit_w=ns)
    trim_kts = [kt for kt in K16_KTS if kt <= 16]
    for kt, ns, zero_steps in itertools.product(trim_kts, (1, 2), (3, 2)):
        _half_steps(ng, oracle, 160, 208, 16 * kt - 4 * zero_steps, seen, split_kernel=-1, nsplit_h=ns, nsplit_w=ns)
    want = set()
    for kt in K16_KTS:
        occ = "2" if kt <= 16 else "1"
        for w, p, d in itertools.product(("false", "true"), ("false", "true"), ("0",)):
            want.add(("fused_step_kernel_k16", (str(kt), w, p, d, "false", occ, "false", "0")))
            if kt in trim_kts:
                if kt > 1:
                    want.add(("fused_step_kernel_k16", (str(kt), w, p, d, "false", occ, "false", "3")))
                want.add(("fused_step_kernel_k16", (str(kt), w, p, d, "false", occ, "false", "2")))
        want.add(("fused_step_kernel_k16", (str(kt), "false", "false", "0", "true", occ, "false", "0")))
    assert want <= seen, sorted(want - seen)
    assert not [a for n, a in seen if n == "fused_step_kernel_k16" and a[3] != "0"]
    before = set(seen)
    _half_steps(ng, oracle, 160, 208, 64, seen, split_kernel=-1, fast_divide=1)     # accepted, ignored: nothing new is launched
    assert seen == before


def test_a_logical_k_between_two_instantiations_runs_the_next_multiple_of_16(ng, oracle, recording):
    """K = 100 computes on 112 (KT = 7) in factors padded to 128; K = 200 on 208; K = 33 on 48; K = 270 on 272, 300 on 304, 400 on 400 (factors padded to 288 / 320 / 416): the reference pads to 32 and nothing coarser (cuda/matrix.cuh:7)."""
    seen = set()
    for K, kt in ((100, 7), (200, 13), (33, 3), (270, 17), (300, 19), (400, 25), (97, 7), (250, 16), (30, 2), (17, 2), (10, 1)):
        seen.clear()
        _half_steps(ng, oracle, 160, 208, K, seen, split_kernel=-1)
        assert {a[0] for n, a in seen if n == "fused_step_kernel_k16"} == {str(kt)}, (K, seen)
        seen.clear()
        if K <= 256:
            _half_steps(ng, oracle, 512, 768, K, seen, split_kernel=1)
            assert {a[0] for n, a in seen if n == "split_step_kernel_k16"} == {str(max(kt, 2))}, (K, seen)   # the split kernel starts at K = 32


def test_every_instantiation_of_the_split_kernel(ng, oracle, recording):
    """split_step_kernel_k16<KT, NW, WSTEP, PARTIAL, DIV, OCC, DB>: K = 32, 48, 64 (two LDS images, two workgroups per CU; K = 64 also
    with eight waves), 80 .. 128 (two images at one workgroup per CU; one image at two per CU), 144 .. 256 (one image); both
    half-steps, both epilogues"""
    seen = set()
    combos = [(32, {}, ("2", "4", "2", "true")), (48, {}, ("3", "4", "2", "true")), (64, {}, ("4", "4", "2", "true")), (64, {"NMF_SPLIT_NW": "8"}, ("4", "8", "2", "true"))]
    for kt in (5, 6, 7, 8):
        combos.append((16 * kt, {}, (str(kt), "4", "1", "true")))
        combos.append((16 * kt, {"NMF_SPLIT_SINGLE": "1"}, (str(kt), "4", "2", "false")))
    for kt in range(9, 17):
        combos.append((16 * kt, {}, (str(kt), "4", "1", "false")))
    for (K, env, _), ns in itertools.product(combos, (1, 2)):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            _half_steps(ng, oracle, 512, 768, K, seen, split_kernel=1, nsplit_h=ns, nsplit_w=ns)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    want = set()
    for (_, _, (kt, nw, occ, db)) in combos:
        for w, p, d in itertools.product(("false", "true"), ("false", "true"), ("0",)):
            want.add(("split_step_kernel_k16", (kt, nw, w, p, d, occ, db)))
    assert want <= seen, sorted(want - seen)


def test_every_instantiation_of_the_wave_pair_kernel(ng, oracle, recording):
    """fused_step_kernel_pair<KTH, WSTEP, PARTIAL, DIV, CHECK> for every K = 32 KTH from 608 to 1024 (round 5: until then K = 640, 768,
    896, 1024 only, and K = 520 ran padded to 640; K <= 576 now stays on the 64-column kernel).  Odd KTH: a wave's half of K ends in a
    remainder block of product 1 (runs of 4, 8 or 12 per lane group), the factors are padded to the next multiple of 64 in HBM and the
    LDS image holds two ds_writes per thread more than product 2 has slots; the slab epilogue writes the 32 padding rows as zeros.
    (The 32-column kernel fused_step_kernel_v3 is reachable only through NMF_FUSED_VARIANT, an A/B switch read once per process, since
    round 4 gave K <= 32 to the 64-column kernel: tests/test_gpu_update_div.py::test_32_column_kernel_family_via_env_override runs it
    in a subprocess.)"""
    seen = set()
    kths = tuple(range(19, 33))
    for kth, ns in itertools.product(kths, (1, 2)):
        _half_steps(ng, oracle, 96, 160, 32 * kth, seen, nsplit_h=ns, nsplit_w=ns)
    want = set()
    for kth in kths:
        for w, p in itertools.product(("false", "true"), ("false", "true")):
            want.add(("fused_step_kernel_pair", (str(kth), w, p, "0", "false")))
        want.add(("fused_step_kernel_pair", (str(kth), "false", "false", "0", "true")))
    got_pair = {(n, a[:5]) for n, a in seen if n == "fused_step_kernel_pair"}
    assert want <= got_pair, sorted(want - got_pair)


def test_a_logical_k_above_512_runs_at_the_reference_granularity(ng, oracle, recording):
    """K = 520 computes on 528 (the 64-column kernel, KT = 33, factors padded to 544) and 570 on 576; 600 on 608 (the wave-pair kernel,
    KTH = 19, factors padded to 640), 700 on 704, 900 on 928 (padded to 960), 1000 on 1024: the reference's granularity (PAD_MULT = 32,
    cuda/matrix.cuh:7; cuda/matrix.cu:88-95) or finer.  With two slabs per half-step too, so that the zero rows the slab epilogue
    writes for the padding are read by the apply kernels."""
    seen = set()
    for K, fam, t in ((520, "fused_step_kernel_k16", 33), (570, "fused_step_kernel_k16", 36), (600, "fused_step_kernel_pair", 19), (700, "fused_step_kernel_pair", 22),
                      (900, "fused_step_kernel_pair", 29), (1000, "fused_step_kernel_pair", 32), (577, "fused_step_kernel_pair", 19)):
        for ns in (1, 2):
            seen.clear()
            _half_steps(ng, oracle, 96, 160, K, seen, nsplit_h=ns, nsplit_w=ns)
            assert {(n, a[0]) for n, a in seen} == {(fam, str(t))}, (K, seen)
