#!/bin/bash
# after removing small launches from the training step: the step's tests, the operator census, the step's time (graphed / eager)
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_transforms_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_gpu.py -m gpu -x -q -k "wave_frontend or c3 or train or fullsize or full_size or engine or ddp or graph or agc or device_draw or datasets" > $out/pytest_small.log 2>&1
rc=$?; tail -3 $out/pytest_small.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python3 scripts/gpu_op_census.py > $out/op_census.log 2>&1; head -3 $out/op_census.log | tail -2
for r in 1 2 3; do timeout -k 10 200 python3 scripts/gpu_graph_train.py 2>&1 | grep -i "graph\|eager" | tail -2; done
