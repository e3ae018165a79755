#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc.sh (gpurun_out/pmc/p*/ ... counter_collection.csv)
into profiles/r6/pmc_traffic.json, stamped with the git sha and the hash of the kernel sources the passes
ran on (bench.py refuses the figure when the sources have changed since).
Corrections per MI355X_MICROARCH.md, section HBM: FETCH_SIZE (KB) reads 1/2 of a wide coalesced read stream on
gfx950 -> doubled; WRITE_SIZE (KB) as is.   usage: python scripts/pmc_summarise.py [pmc_dir] [out_json]"""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha, ALGO_BYTES_PER_AUDIO_S, BATCH, SECONDS  # noqa: E402

pmc_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r5", "pmc_traffic.json")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(pmc_dir, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))


def short(name):
    name = name.replace("void ", "").replace("(FusedArgs)", "").replace(" ", "")
    return name.split("(")[0]


doc = {
    "_how": "scripts/gpu_pmc.sh: rocprofv3 --kernel-trace --pmc <group> (one pass per group, no other trace domain) around "
            "`python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-extras` (steps rotate "
            "through 20 distinct c2 batches = 512 MB per cycle, so reads come from HBM, not the Infinity Cache); "
            "mean over the dispatches of each kernel; MI355X gfx950, ROCm 7.2",
    "_correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) x2 on gfx950 for wide coalesced reads; WRITE_SIZE (KB) as is",
    "git_sha": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
    "kernel_src_sha": kernel_source_sha(),
}
for k, v in agg.items():
    if "k_wav_to_mel" not in k and "k_minmax" not in k:
        continue
    rec = {c: round(sum(x) / len(x), 1) for c, x in v.items()}
    rec["dispatches"] = len(next(iter(v.values())))
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_read_bytes_corrected"] = int(rec["FETCH_SIZE"] * 1024 * 2)
        rec["hbm_write_bytes"] = int(rec["WRITE_SIZE"] * 1024)
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_corrected"] + rec["hbm_write_bytes"]
        if "k_wav_to_mel" in k:
            rec["algorithmic_bytes_per_launch"] = ALGO_BYTES_PER_AUDIO_S * BATCH * SECONDS
            rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], 3)
    doc[short(k)] = rec
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
