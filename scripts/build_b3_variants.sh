#!/bin/bash
# Ablation builds of the bf16x3 Winograd kernel for the microbenchmark (timing only: results are wrong when a phase is skipped).
cd "$(dirname "$0")/.."
for abl in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -DIRIS_B3_ABLATE=$abl -o scripts/microbench/libwino_abl$abl.so scripts/microbench/wino_conv.hip &
done
wait
ls -la scripts/microbench/libwino_abl*.so
