#!/bin/bash
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_transforms_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_gpu.py -m gpu -x -q -k "packings or fullsize or full_size or train or ddp or graph" > $out/pytest_pack.log 2>&1
rc=$?; tail -4 $out/pytest_pack.log
if [ $rc -ne 0 ]; then exit $rc; fi
scripts/gpu_r6_step_ab.sh IRIS_FUSED_PACK 3 > $out/c4_fused_pack_ab.log 2>&1; cat $out/c4_fused_pack_ab.log
