"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per counter group) into markdown tables.
    python tools/pmc_summary.py [--json profiles/pmc_static.json --shape 4096x65536x256 --source "..."] DIR [DIR ...]
Per kernel name: mean counter value per dispatch (and mean duration from the dispatch timestamps).
--json: also merge {shape: {hbm_bytes_per_launch: {H, W}, mfma_busy_frac_of_simd_cycles: {H, W}, source}} for the half-step kernels of
this run into the given file -- what bench.py reports as roofline.traffic (it does not collect counters itself)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

opt = {}
argv = sys.argv[1:]
while argv and argv[0].startswith("--"):
    opt[argv[0][2:]] = argv[1]
    argv = argv[2:]
sys.argv[1:] = argv

val = defaultdict(lambda: defaultdict(list))    # kernel -> counter -> values
dur = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            val[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (f, r["Dispatch_Id"]) not in seen and "SQ_WAVE_CYCLES" == r["Counter_Name"]:
                seen.add((f, r["Dispatch_Id"]))
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
mean = lambda v: sum(v) / len(v) if v else float("nan")
ks = [k for k in val if "nmf::" in k]
print("| kernel | FETCH_SIZE raw KB | WRITE_SIZE KB | HBM bytes per launch = (2*FETCH + WRITE)*1024 |\n|---|---|---|---|")
for k in sorted(ks):
    c = val[k]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fe, wr = mean(c["FETCH_SIZE"]), mean(c["WRITE_SIZE"])
        print(f"| `{k.split('(')[0]}` | {fe:.0f} | {wr:.0f} | {(2 * fe + wr) * 1024 / 1e9:.3f} GB |")
print()
for k in sorted(ks):
    c = val[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "step_kernel" not in k:
        continue
    g, w = mean(c["GRBM_GUI_ACTIVE"]) / 8, mean(c["SQ_WAVE_CYCLES"])     # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    ms = mean(dur[k])
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD and is summed over the chip: 256 CUs x 4 SIMDs x kernel cycles = 100 %
    print(f"* `{k.split('(')[0]}`: {ms:.3f} ms under the profiler, clock {g / ms / 1e6:.3f} GHz; SQ_VALU_MFMA_BUSY_CYCLES {mean(c['SQ_VALU_MFMA_BUSY_CYCLES']):.3e}"
          f" = **{100 * mean(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (g * 1024):.1f} % of SIMD-cycles**; of wave-cycles: SQ_WAIT_ANY {100 * mean(c['SQ_WAIT_ANY']) / w:.1f} %,"
          f" SQ_WAIT_INST_ANY {100 * mean(c['SQ_WAIT_INST_ANY']) / w:.1f} %, SQ_ACTIVE_INST_ANY {100 * mean(c['SQ_ACTIVE_INST_ANY']) / w:.1f} %,"
          f" SQ_ACTIVE_INST_VALU {100 * mean(c['SQ_ACTIVE_INST_VALU']) / w:.1f} %; SQ_LDS_BANK_CONFLICT {mean(c['SQ_LDS_BANK_CONFLICT']):.2e} of SQ_LDS_IDX_ACTIVE {mean(c['SQ_LDS_IDX_ACTIVE']):.2e}")
print()
for k in sorted(ks):
    c = val[k]
    if "SQ_INSTS_MFMA" not in c or "step_kernel" not in k:
        continue
    mf, va = mean(c["SQ_INSTS_MFMA"]), mean(c["SQ_INSTS_VALU"])
    print(f"* `{k.split('(')[0]}`: MFMA {mf:.3e}, VALU incl. MFMA {va:.3e} (non-MFMA VALU per MFMA: {(va - mf) / mf:.2f}), LDS {mean(c['SQ_INSTS_LDS']):.3e},"
          f" VMEM {mean(c['SQ_INSTS_VMEM']):.3e}, SALU {mean(c['SQ_INSTS_SALU']):.3e}")


def half_step(k):
    """'H' / 'W' for a half-step instantiation (CHECK and GEMM modes excluded), else None"""
    m = re.search(r"nmf::(fused_step_kernel_k16|split_step_kernel_k16|fused_step_kernel_pair|fused_step_kernel_v3)<([^>]*)>", k)
    if not m:
        return None
    a = [x.strip() for x in m.group(2).split(",")]
    if m.group(1) == "split_step_kernel_k16":
        return "W" if a[2] == "true" else "H"
    if (m.group(1) == "fused_step_kernel_k16" and (a[4] == "true" or a[6] == "true")) or (m.group(1) == "fused_step_kernel_pair" and a[4] == "true"):
        return None
    if m.group(1) == "fused_step_kernel_v3" and len(a) > 5 and a[5] == "true":
        return None
    return "W" if a[1] == "true" else "H"


if "json" in opt:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kernel_identity import kernel_sources_sha16
    # the passes are of the build in this tree: bench.py quotes them only while the kernel sources are the same (tools/kernel_identity.py)
    entry = {"source": opt.get("source", "rocprofv3 --pmc passes (tools/pmc_summary.py)"), "hbm_bytes_per_launch": {}, "mfma_busy_frac_of_simd_cycles": {}, "kernel": {},
             "kernel_sources_sha16": kernel_sources_sha16()}
    for k in sorted(ks):
        hw, c = half_step(k), val[k]
        if hw and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            entry["hbm_bytes_per_launch"][hw] = (2 * mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"])) * 1024
            entry["kernel"][hw] = k.split("(")[0].replace("void ", "")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                entry["mfma_busy_frac_of_simd_cycles"][hw] = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (mean(c["GRBM_GUI_ACTIVE"]) / 8 * 1024)
    try:
        allv = json.load(open(opt["json"]))
    except Exception:
        allv = {}
    allv[opt["shape"]] = entry
    json.dump(allv, open(opt["json"], "w"), indent=1, sort_keys=True)
