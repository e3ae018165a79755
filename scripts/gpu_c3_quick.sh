cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_transforms_gpu.py -x -q -m gpu -k "bias_relu or inference_engine" 2>&1 | tail -3
for mode in module engine graph; do
  timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 $mode > /dev/null 2>&1
  timeout -k 10 300 python3 scripts/gpu_fwdprof.py 30 $mode 2>&1 | grep "fwd\["
done
