#!/bin/bash
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_transforms_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_gpu.py -m gpu -x -q -k "agc or train or fullsize or full_size or ddp or graph or fit or bench" > $out/pytest_adam_all.log 2>&1
rc=$?; tail -4 $out/pytest_adam_all.log
if [ $rc -ne 0 ]; then exit $rc; fi
scripts/gpu_r6_step_ab.sh IRIS_FUSED_ADAM_AGC 3 > $out/c4_fused_adam_ab.log 2>&1; cat $out/c4_fused_adam_ab.log
