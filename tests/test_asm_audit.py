"""Static guard for the inline-asm MFMAs (tools/asm_audit.py): runs on the CPU, needs hipcc only.  Every kernel translation
unit is compiled to gfx950 assembly with the Makefile's flags; every MFMA inside an `asm volatile` statement is checked for
(a) a VALU write of one of its operands within 2 wait states before it and (b) any non-accumulating touch of its result
within passes + 4 wait states after it, along every control-flow path; kernels holding asm MFMAs must not spill or use
scratch.  The detector itself is first run on hand-written snippets of both hazards and of their padded forms."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_audit  # noqa: E402


def test_the_detector_flags_both_hazards_and_accepts_their_padded_forms():
    assert asm_audit.self_test() == []


@pytest.mark.skipif(not os.path.exists(asm_audit.HIPCC), reason="needs hipcc")
def test_no_inline_asm_mfma_of_the_shipped_kernels_sits_in_a_hazard_window():
    n, hazards, rows = asm_audit.run()
    assert hazards == [], "\n".join(hazards)
    assert n >= 10000                     # the product-1 chains of every instantiation were found and walked (the split kernel's too:
                                          # its early exit for inactive pairs ends in an s_endpgm of its own, ahead of the body)
    kernels = {r[1].split("<")[0] for r in rows}
    for family in ("nmf::fused_step_kernel_k16", "nmf::split_step_kernel_k16", "nmf::fused_step_kernel_pair", "nmf::fused_step_kernel_v3"):
        assert family in kernels, (family, sorted(kernels)[:5])
    with_asm = [r for r in rows if r[9] > 0]
    assert with_asm and all(r[6] == 0 and r[7] == 0 for r in with_asm)      # no scratch, no spills next to asm MFMAs
    # The f32 MFMA shares the VALU datapath: address arithmetic and register copies inside the chunk loop are MFMA issue time.  The production
    # half-step kernels of the 64-column and the wave-pair family keep them to a handful per chunk (round 5: the K = 576 kernel had 81 v_add_u32
    # per chunk, one per LDS access beyond the 64 KiB an immediate offset reaches, and ran at 82 % instead of 90 % of peak).
    worst = {}
    for r in rows:
        fam = r[1].split("<")[0]
        if fam in ("nmf::fused_step_kernel_k16", "nmf::fused_step_kernel_pair") and r[10] >= 16:
            worst[fam] = max(worst.get(fam, 0), r[11])
            assert r[11] <= (0.15 * r[10] if r[10] >= 256 else max(8, 0.35 * r[10])), (r[1], r[10], r[11])
    assert set(worst) == {"nmf::fused_step_kernel_k16", "nmf::fused_step_kernel_pair"}, worst
