#!/bin/bash
# times the bf16x3 Winograd kernel's ablation builds (scripts/build_b3_variants.sh) on two layers
out=gpurun_out/r6; mkdir -p $out
for abl in "$@"; do
  echo "== IRIS_B3_ABLATE=$abl"
  WINO_LIB=scripts/microbench/libwino_abl$abl.so WINO_ROWS=short timeout -k 10 200 python scripts/gpu_wino_b3_bench.py time 2>&1 | grep "pool\|sum"
done
