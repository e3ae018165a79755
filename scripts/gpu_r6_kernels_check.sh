#!/bin/bash
# After a change to the Winograd kernels: the standalone check + timing of both (exact fp32 / split bf16), then their tests.
out=gpurun_out/r6; mkdir -p $out
tag=${1:-k}
timeout -k 10 400 python3 scripts/gpu_wino_b3_bench.py all > $out/wino_b3_bench_$tag.log 2>&1
rc=$?; grep "gate\|sum of" $out/wino_b3_bench_$tag.log
if [ $rc -ge 124 ]; then echo "killed at its limit"; exit $rc; fi
timeout -k 10 600 python -m pytest tests/test_transforms_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "wino or fullsize or full_size or batchnorm_statistics or engine or train" > $out/pytest_kernels_$tag.log 2>&1
rc2=$?; tail -3 $out/pytest_kernels_$tag.log
exit $(( rc > rc2 ? rc : rc2 ))
