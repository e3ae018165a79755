#!/bin/bash
# A/B of Winograd kernel builds in one session: scripts/microbench/libwino_<tag>.so for every tag given, three alternating rounds of
# the twelve-layer timing (scripts/gpu_wino_b3_bench.py time).   usage: gpu_wino_ab.sh <tagA> <tagB> ...
out=gpurun_out/r6; mkdir -p $out
for r in 1 2 3; do
  for tag in "$@"; do
    WINO_LIB=scripts/microbench/libwino_$tag.so timeout -k 10 200 python3 scripts/gpu_wino_b3_bench.py time > $out/wino_ab_${tag}_$r.log 2>&1
    rc=$?
    if [ $rc -ge 124 ]; then echo "$tag round $r killed at its limit"; exit $rc; fi
    echo "$tag round $r: $(grep 'sum of' $out/wino_ab_${tag}_$r.log)"
  done
done
