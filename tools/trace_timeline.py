"""Timeline of the last dispatches in a rocprofv3 --kernel-trace CSV: per kernel its duration and the gap since the
previous kernel ended, then per-kernel-name totals over that window.
    python tools/trace_timeline.py <dir-or-csv> [n_last]"""
import csv, glob, os, sys
from collections import defaultdict
p = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 64
files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void nmf::", "").replace("nmf::", ""), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
rows.sort()
rows = rows[-n_last:]
tot = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for i, (s, e, name, grid, wg) in enumerate(rows):
    gap = (s - prev_end) * 1e-3 if prev_end is not None else 0.0
    if i < 24: print(f"{name[:70]:70s} grid {grid:>8s} wg {wg:>5s}  dur {(e - s) * 1e-3:8.2f} us  gap {gap:7.2f} us")
    t = tot[name]; t[0] += 1; t[1] += (e - s) * 1e-3; t[2] += gap
    prev_end = e
span = (rows[-1][1] - rows[0][0]) * 1e-3
print(f"--- window: {len(rows)} dispatches, {span:.1f} us")
for name, (n, d, g) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:70]:70s} n {n:4d}  dur/launch {d / n:8.2f} us  gap-before/launch {g / n:6.2f} us  share {100 * (d + g) / span:5.1f} %")
