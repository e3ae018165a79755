#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch table from the device assembly (make -C challenge_amd/csrc asm).
usage: python scripts/kernel_resources.py [iris_frontend.s] [name-filter]"""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "challenge_amd/csrc/iris_frontend.s"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name, rows = None, []
for line in open(path):
    m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
    if m:
        name, cur = m.group(1), {}
        continue
    if name:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|accum_offset)\s+(\d+)", line)
        if m:
            cur[m.group(1)] = int(m.group(2))
        if ".end_amdhsa_kernel" in line:
            rows.append((name, cur)); name = None
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (n, c), d in zip(rows, names):
    d = d.replace("(FusedArgs)", "").replace("void ", "")
    if flt in d:
        v = c.get("next_free_vgpr", 0); alloc = (v + 7) // 8 * 8
        print(f"{d:60s} vgpr {v:4d} (alloc {alloc:3d}, {min(8, 512 // max(alloc,1))} waves/SIMD)  sgpr {c.get('next_free_sgpr',0):3d}  scratch {c.get('private_segment_fixed_size',0)}")
