#!/usr/bin/env python3
"""Build-time guard for the inline-asm MFMAs.

The product-1 chains of the fused kernels are `asm volatile("v_mfma_f32_16x16x4_f32 ...")` statements (so that their
result lives in VGPRs and source order is issue order).  hipcc treats an asm statement as one opaque instruction: it pads
none of the hazards of the MFMA inside (cdna_hip_programming.md 5.7 item 2).  Two of them matter here:

  (a) a VALU write (v_mov, v_accvgpr_read, ...) of a register the asm MFMA reads as A, B or C needs 2 wait states before
      the MFMA -- the miscompile of round 2 (K = 256 split kernel: the register allocator parked operands in AGPRs and
      copied them back right in front of the asm MFMA; ~1 % wrong sums, no fault);
  (b) the MFMA's result needs passes + 4 wait states (12 for 16x16x4 f32, 20 for 32x32x2 f32) before anything but an
      accumulating MFMA touches it -- the kernels end every asm chain with `s_nop 15; s_nop 3` for that.

This script compiles every kernel translation unit to gfx950 assembly (hipcc -S, the Makefile's flags), walks every
kernel, and for each MFMA between ;;#ASMSTART / ;;#ASMEND checks (a) backwards and (b) forwards through the control-flow
graph.  It also tabulates VGPR / AGPR / SGPR / LDS / scratch / spill counts of every kernel (profiles/r03_kernel_resources.md)
and fails on a spill or scratch use in a kernel that contains asm MFMAs.  Run by tests/test_asm_audit.py (CPU, no GPU needed).

  python tools/asm_audit.py [--out profiles/r03_kernel_resources.md] [--keep DIR]
"""
import argparse
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmf-gpu_amd", "csrc")
# (label, source, extra flags): the instantiations of the 64-column, the split and the wave-pair kernel are one source each, compiled in
# four groups (csrc/Makefile)
UNITS = [(f"nmf_fused16_inst{g}", "nmf_fused16_inst", (f"-DNMF_K16_GROUP={g}",)) for g in range(4)] + \
        [(f"nmf_split16_inst{g}", "nmf_split16_inst", (f"-DNMF_S16_GROUP={g}",)) for g in range(4)] + \
        [(f"nmf_pair16_inst{g}", "nmf_pair16_inst", (f"-DNMF_P16_GROUP={g}",)) for g in range(4)] + \
        [(u, u, ()) for u in ("nmf_fused16", "nmf_pair16", "nmf_split16", "nmf_fused32", "nmf_kernels", "nmf_gemm")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950",
         "-mllvm", "-enable-misched=false", "-mllvm", "-pragma-unroll-threshold=1000000"]   # csrc/Makefile: CXXFLAGS + KFLAGS
OPERAND_WAIT = 2          # VALU write -> MFMA A/B/C read
RESULT_WAIT = {"16x16x4": 12, "32x32x2": 20}   # passes + 4


def compile_unit(unit, outdir):
    label, source, extra = unit
    out = os.path.join(outdir, label + ".s")
    subprocess.run([HIPCC] + FLAGS + list(extra) + ["--cuda-device-only", "-S", os.path.join(CSRC, source + ".hip"), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def self_test():
    """The detector on hand-written snippets of the two hazards (and their padded, legal forms): the round-2 miscompile pattern
    -- a compiler `v_accvgpr_read` of an AGPR-parked operand directly in front of an asm MFMA that reads it -- must be flagged,
    also when the copy sits at the end of a loop body that branches back to the MFMA; `s_nop 1` between them clears it; a VALU
    read of the result right after the chain must be flagged, the accumulate chain itself and a read behind
    `s_nop 15; s_nop 3` must not."""
    def kernel(body):
        return "_Z4testv:\n" + body + "\n\ts_endpgm\n"
    asm = lambda t: "\t;;#ASMSTART\n\t" + t + "\n\t;;#ASMEND\n"
    cases = {
        "copy_before_asm_mfma": (kernel("\tv_accvgpr_read_b32 v5, a3\n" + asm("v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]") + asm("s_nop 15\n\ts_nop 3")), 1),
        "copy_padded": (kernel("\tv_accvgpr_read_b32 v5, a3\n\ts_nop 1\n" + asm("v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]") + asm("s_nop 15\n\ts_nop 3")), 0),
        "copy_on_back_edge": (kernel(".LBB0_1:\n" + asm("v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]") + asm("s_nop 15\n\ts_nop 3")
                                     + "\ts_add_u32 s0, s0, 1\n\tv_mov_b32_e32 v4, v9\n\ts_cbranch_scc1 .LBB0_1\n"), 1),
        "result_read_too_early": (kernel(asm("v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, 0") + asm("v_mfma_f32_16x16x4_f32 v[0:3], v6, v7, v[0:3]")
                                         + "\ts_nop 7\n\tv_max_f32_e32 v8, v0, v9\n"), 1),
        "result_read_after_nops": (kernel(asm("v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, 0") + asm("v_mfma_f32_16x16x4_f32 v[0:3], v6, v7, v[0:3]")
                                          + asm("s_nop 15\n\ts_nop 3") + "\tv_max_f32_e32 v8, v0, v9\n"), 0),
        "compiler_mfma_is_not_audited": (kernel("\tv_accvgpr_read_b32 v5, a3\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\tv_max_f32_e32 v8, v0, v9\n"), 0),
    }
    bad = []
    with tempfile.TemporaryDirectory() as d:
        for name, (text, want) in cases.items():
            path = os.path.join(d, name + ".s")
            open(path, "w").write(text)
            got = sum(len(audit_kernel(k, c)[1]) for k, c in parse_kernels(path).items())
            if (got > 0) != (want > 0):
                bad.append(f"{name}: {got} hazards reported, expected {'some' if want else 'none'}")
    return bad


_REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(text):
    """set of ('v'|'a', index) named in an operand string"""
    out = set()
    for m in _REG.finditer(text):
        if m.group(2) is not None:
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(1), int(m.group(4))))
    return out


class Ins:
    __slots__ = ("op", "ops", "line", "in_asm", "labels")

    def __init__(self, op, ops, line, in_asm):
        self.op, self.ops, self.line, self.in_asm, self.labels = op, ops, line, in_asm, []

    def states(self):
        if self.op == "s_nop":
            return int(self.ops[0], 0) + 1
        return 1

    def is_mfma(self):
        return self.op.startswith("v_mfma")

    def writes(self):
        """vector registers written (asynchronous loads excluded: the compiler waits for those with s_waitcnt)"""
        if not self.op.startswith("v_") or self.op.startswith(("v_cmp", "v_nop", "v_readfirstlane", "v_readlane")) or not self.ops:
            return set()
        return regs(self.ops[0])

    def touches(self):
        return set().union(*[regs(o) for o in self.ops]) if self.ops else set()


def parse_kernels(path):
    """{kernel name: [Ins]}; labels attached to the instruction they precede"""
    kernels, cur, name, in_asm, pending = {}, None, None, False, []
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";;#")[0] if ";;#" in raw else raw
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        text = line.split(";")[0].strip()
        if not text:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):$", text)
        if m:
            lab = m.group(1)
            if lab.startswith("_Z") and cur is None:
                name, cur, pending = lab, [], []
            elif lab.startswith(".Lfunc_end") and cur is not None:
                # the end of the function, not the first s_endpgm: a kernel with an early exit (the split kernel's inactive
                # pairs) has several, and its main body follows the first
                kernels[name] = cur
                cur, name = None, None
            elif cur is not None:
                pending.append(lab)
            continue
        if text.startswith("."):
            if text.startswith(".end_amdhsa_kernel") or text.startswith(".Lfunc_end"):
                pass
            continue
        if cur is None:
            continue
        parts = text.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        # register ranges contain no commas (v[0:3]), so the split is safe
        ins = Ins(op, ops, ln, in_asm)
        ins.labels, pending = pending, []
        cur.append(ins)
    if cur is not None:      # hand-written snippets (self_test) carry no .Lfunc_end label
        kernels[name] = cur
    return kernels


def audit_kernel(name, code):
    """list of hazard descriptions for the asm MFMAs of one kernel"""
    label_at = {}
    for i, ins in enumerate(code):
        for lab in ins.labels:
            label_at[lab] = i
    branches_to = {}
    for i, ins in enumerate(code):
        if ins.op.startswith(("s_branch", "s_cbranch")) and ins.ops and ins.ops[0] in label_at:
            branches_to.setdefault(label_at[ins.ops[0]], []).append(i)
    hazards = []

    def preds(i):
        p = []
        if i > 0 and code[i - 1].op != "s_branch":
            p.append(i - 1)
        if code[i].labels:
            p.extend(branches_to.get(i, []))
        return p

    def succs(i):
        ins = code[i]
        if ins.op == "s_endpgm":
            return []
        s = []
        if ins.op.startswith(("s_branch", "s_cbranch")) and ins.ops and ins.ops[0] in label_at:
            s.append(label_at[ins.ops[0]])
        if ins.op != "s_branch" and i + 1 < len(code):
            s.append(i + 1)
        return s

    n_asm = 0
    for i, ins in enumerate(code):
        if not (ins.in_asm and ins.is_mfma()):
            continue
        n_asm += 1
        dst = regs(ins.ops[0])
        srcs = set().union(*[regs(o) for o in ins.ops[1:]])
        shape = "32x32x2" if "32x32x2" in ins.op else "16x16x4"
        # (a) backwards: VALU writers of a source within OPERAND_WAIT states
        seen, stack = set(), [(p, OPERAND_WAIT) for p in preds(i)]
        while stack:
            j, left = stack.pop()
            if left <= 0 or (j, left) in seen:
                continue
            seen.add((j, left))
            w = code[j]
            if not w.is_mfma() and (w.writes() & srcs):
                hazards.append(f"{name}: line {ins.line}: asm {ins.op} reads {sorted(w.writes() & srcs)} written by `{w.op} {', '.join(w.ops)}` "
                               f"(line {w.line}) {OPERAND_WAIT - left} wait state(s) earlier; {OPERAND_WAIT} needed")
            for p in preds(j):
                stack.append((p, left - w.states()))
        # (b) forwards: anything but an accumulating MFMA touching the result within RESULT_WAIT states
        need = RESULT_WAIT[shape]
        seen, stack = set(), [(s, need) for s in succs(i)]
        while stack:
            j, left = stack.pop()
            if left <= 0 or (j, left) in seen:
                continue
            seen.add((j, left))
            r = code[j]
            if r.op == "s_endpgm":
                continue
            if r.touches() & dst:
                chain = r.is_mfma() and regs(r.ops[0]) == dst and len(r.ops) > 3 and regs(r.ops[3]) == dst and not (set().union(regs(r.ops[1]), regs(r.ops[2])) & dst)
                if chain:
                    continue       # accumulate chain: the next MFMA takes D whole as C (0 states); it starts a window of its own
                hazards.append(f"{name}: line {ins.line}: result {ins.ops[0]} of asm {ins.op} is touched by `{r.op} {', '.join(r.ops)}` (line {r.line}) "
                               f"after {need - left} wait state(s); {need} needed")
                continue
            for s in succs(j):
                stack.append((s, left - r.states()))
    return n_asm, hazards


ADDR_COPY_OPS = ("v_add_u32", "v_add3_u32", "v_lshl_add_u32", "v_mad_u32_u24", "v_accvgpr_read", "v_accvgpr_write", "v_mov_b32")


def chunk_loop_overhead(code):
    """(MFMAs, address-arithmetic / register-copy VALU instructions) inside the kernel's longest backward-branch region -- the chunk loop of
    the half-step kernels.  The f32 MFMA shares the VALU datapath, so every such instruction is MFMA issue time; round 5 found 81 of them per
    chunk in the K = 576 kernel (one v_add_u32 per LDS access beyond the 64 KiB an immediate offset reaches: 82 % instead of 90 % of peak)."""
    label_at = {}
    for i, ins in enumerate(code):
        for lab in ins.labels:
            label_at[lab] = i
    best = None
    for i, ins in enumerate(code):
        if ins.op.startswith(("s_cbranch", "s_branch")) and ins.ops and ins.ops[0] in label_at and label_at[ins.ops[0]] < i:
            span = (label_at[ins.ops[0]], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        return 0, 0
    loop = code[best[0]:best[1]]
    return sum(1 for c in loop if c.is_mfma()), sum(1 for c in loop if c.op.startswith(ADDR_COPY_OPS))


def resources(path):
    """per-kernel register / LDS / scratch figures from the amdhsa metadata at the end of the .s"""
    out, cur = {}, None
    keys = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size",
            ".vgpr_spill_count", ".sgpr_spill_count", ".max_flat_workgroup_size")
    for line in open(path):
        t = line.strip()
        if t.startswith("- .agpr_count:") or t.startswith("- .args:"):
            cur = {}
        m = re.match(r"^-?\s*(\.[a-z_]+):\s*(.+)$", t)
        if m and cur is not None and m.group(1) in keys:
            cur[m.group(1)] = m.group(2).strip()
            if m.group(1) == ".name":
                out[cur[".name"]] = cur
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.split("\n")))
    except Exception:
        return {n: n for n in names}


def run(outdir=None, report=None):
    tmp = None
    if outdir is None:
        tmp = tempfile.TemporaryDirectory()
        outdir = tmp.name
    os.makedirs(outdir, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        paths = list(ex.map(lambda u: compile_unit(u, outdir), UNITS))
    all_hazards, rows, n_asm_total = [], [], 0
    for (unit, _, _), path in zip(UNITS, paths):
        kernels = parse_kernels(path)
        res = resources(path)
        pretty = demangle(list(kernels))
        for name, code in kernels.items():
            n_asm, hz = audit_kernel(pretty[name].split("(")[0], code)
            n_asm_total += n_asm
            all_hazards += hz
            r = res.get(name, {})
            spill = int(r.get(".vgpr_spill_count", 0)) + int(r.get(".sgpr_spill_count", 0))
            scratch = int(r.get(".private_segment_fixed_size", 0))
            if n_asm and (spill or scratch):
                all_hazards.append(f"{pretty[name]}: {spill} spilled registers, {scratch} bytes of scratch in a kernel with inline-asm MFMAs")
            n_mfma = sum(1 for c in code if c.is_mfma())
            loop_mfma, loop_over = chunk_loop_overhead(code)
            rows.append((unit, pretty[name].split("(")[0].replace("void ", ""), r.get(".vgpr_count", "?"), r.get(".agpr_count", "?"), r.get(".sgpr_count", "?"),
                         r.get(".group_segment_fixed_size", "?"), scratch, spill, n_mfma, n_asm, loop_mfma, loop_over))
    if report:
        with open(report, "w") as f:
            f.write("# kernel resources and inline-asm MFMA audit (tools/asm_audit.py; hipcc -S with the Makefile's flags, gfx950)\n\n")
            f.write("VGPR / AGPR / SGPR counts, static LDS bytes (the fused kernels add dynamic LDS at launch), scratch bytes, spilled registers, MFMA "
                    "instructions in the kernel's code, how many of those sit inside `asm volatile` statements, and -- for the longest loop of the kernel, the chunk loop of "
                    "the half-step kernels -- its MFMAs next to its address-arithmetic / register-copy VALU instructions (v_add_u32, v_add3, v_lshl_add, v_mad_u32_u24, "
                    "v_accvgpr_*, v_mov: MFMA issue time on this datapath).  Every asm MFMA was checked for a VALU "
                    f"write of one of its operands within {OPERAND_WAIT} wait states before it and for any non-accumulating touch of its result within "
                    f"{RESULT_WAIT['16x16x4']} (16x16x4) / {RESULT_WAIT['32x32x2']} (32x32x2) wait states after it, along every control-flow path.\n\n")
            f.write(f"**{n_asm_total} inline-asm MFMAs audited, {len(all_hazards)} hazards.**\n\n")
            f.write("| unit | kernel | VGPR | AGPR | SGPR | static LDS | scratch | spills | MFMAs | in asm | loop MFMAs | loop addr/copy VALU |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
            for row in rows:
                f.write("| " + " | ".join(str(c) for c in row) + " |\n")
            if all_hazards:
                f.write("\n## hazards\n\n" + "\n".join("* " + h for h in all_hazards) + "\n")
    if tmp:
        tmp.cleanup()
    return n_asm_total, all_hazards, rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None, help="write the resource / audit table here (markdown)")
    ap.add_argument("--keep", default=None, help="keep the generated .s files in this directory")
    a = ap.parse_args()
    n, hz, rows = run(a.keep, a.out)
    print(f"{n} inline-asm MFMAs in {len(rows)} kernels audited, {len(hz)} hazards")
    for h in hz:
        print("HAZARD:", h)
    sys.exit(1 if hz else 0)
