cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_transforms_gpu.py -x -q -m gpu -k "mixed_layout or train_step or inference_engine or sj_train_main" 2>&1 | tail -3
for env in "IRIS_ALL_NHWC=1" "IRIS_ALL_NHWC="; do
  env $env timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 module > /dev/null 2>&1
  echo "$env"; env $env timeout -k 10 300 python3 scripts/gpu_fwdprof.py 30 module 2>&1 | grep "fwd\["
done
timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > /dev/null 2>&1
timeout -k 10 300 python3 scripts/gpu_c4prof.py 20 2>&1 | grep "train step"
timeout -k 10 300 python3 scripts/gpu_c4prof.py 20 2>&1 | grep "train step"
