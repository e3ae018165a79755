#!/usr/bin/env python3
"""Per-step kernel statistics from a rocprofv3 --kernel-trace CSV: keeps only the dispatches of the LAST n steps, a step
being everything from one dispatch of the marker kernel (the fused frontend kernel, one per step) to the next - so that
warm-up work (MIOpen's find step benchmarks every solver once per shape) stays out of the figures.  Writes a CSV with
the columns of rocprofv3's *_kernel_stats.csv (Calls and TotalDurationNs summed over the kept steps) and prints the
step span (first dispatch start -> last dispatch end, mean over the kept steps) and the busy time inside it.
usage: python scripts/trace_steps.py <kernel_trace.csv> <marker substring> <n_last_steps> <out.csv>"""
import csv, statistics, sys

path, marker, n_last, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(marks) > n_last, f"only {len(marks)} marker dispatches"
first = marks[-n_last - 1]   # the last step may be incomplete at its tail: use the n steps before the final marker
last = marks[-1]
kept = rows[first:last]
agg = {}
for s, e, name in kept:
    agg.setdefault(name, []).append(e - s)
total = sum(sum(v) for v in agg.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / total, 2), min(v), max(v),
                    statistics.pstdev(v) if len(v) > 1 else 0.0])
spans = [(rows[marks[-n_last - 1 + i + 1]][0] - rows[marks[-n_last - 1 + i]][0]) for i in range(n_last)]
print(f"{n_last} steps kept: {len(kept)} dispatches, {len(kept) / n_last:.1f} per step; step period {statistics.mean(spans) / 1e6:.3f} ms, "
      f"kernel busy time {total / n_last / 1e6:.3f} ms per step")
