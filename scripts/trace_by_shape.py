#!/usr/bin/env python3
"""Mean duration of each kernel per consecutive run of identical (name, grid) dispatches in a rocprofv3 kernel trace:
   python scripts/trace_by_shape.py <kernel_trace.csv> <name filter regex> [min run length]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
min_run = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = [r for r in rows if pat.search(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = {}
order = []
# a kernel whose grid does not depend on the shape (k_wino_wrw) is labelled with the grid of the dispatch that follows it
nxt = [rows[i + 1].get("Grid_Size_X", "?") if i + 1 < len(rows) else "?" for i in range(len(rows))]
for r, n in zip(rows, nxt):
    key = (r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")) + "/" + n, r.get("Workgroup_Size_X", "?"))
    if key not in runs:
        runs[key] = []
        order.append(key)
    runs[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key in order:
    d = runs[key]
    if len(d) >= min_run:
        d2 = sorted(d)[: max(1, len(d) * 3 // 4)]
        print(f"{key[0]:60s} grid/next {key[1]:>16s} wg {key[2]:>5s} n {len(d):4d} mean {sum(d) / len(d) / 1e3:8.1f} us  (fastest 3/4: {sum(d2) / len(d2) / 1e3:8.1f})")
