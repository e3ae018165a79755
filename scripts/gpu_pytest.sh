#!/bin/bash
# On the GPU box: the whole GPU test suite (no -x: every failure shows), log under gpurun_out/r6/.
out=gpurun_out/r6; mkdir -p $out
tag=${1:-run}; shift
timeout -k 10 1100 python -m pytest tests -m gpu -q -s "$@" > $out/pytest_gpu_$tag.log 2>&1
rc=$?; tail -12 $out/pytest_gpu_$tag.log; exit $rc
