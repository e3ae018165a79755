#!/usr/bin/env python3
"""Compact view of a rocprofv3 *_kernel_stats.csv: short kernel names, calls, average, total, share.
usage: python scripts/kstats.py <kernel_stats.csv> [top_n] [calls_divisor]"""
import csv, re, sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    if name.startswith("_ZN2ck"):
        m = re.search(r"kernel_(\w+?)INS", name)
        return "ck::" + (m.group(1) if m else "kernel")
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "tensile_gemm_" + (m.group(1) if m else "") + ("_bias" if "_Bias_" in name else "")
    name = re.sub(r"at::native::(\(anonymous namespace\)::)?", "", name)
    name = re.sub(r"<.*", "", name) if len(name) > 90 else name
    return name[:90]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':<92}{'calls':>8}{'avg us':>10}{'total ms':>10}{'share':>8}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print(f"{short(r['Name']):<92}{float(r['Calls']) / div:>8.1f}{float(r['AverageNs']) / 1e3:>10.1f}"
              f"{float(r['TotalDurationNs']) / 1e6 / div:>10.3f}{100 * float(r['TotalDurationNs']) / total:>7.1f}%")
    print(f"{'all kernels':<92}{sum(float(r['Calls']) for r in rows) / div:>8.1f}{'':>10}{total / 1e6 / div:>10.3f}")


if __name__ == "__main__":
    main()
