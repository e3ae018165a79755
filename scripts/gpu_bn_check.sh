cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_transforms_gpu.py -x -q -m gpu -k "fused_bn or train_step or sj_train_main or fused_agc" 2>&1 | tail -12
for v in 0 1; do
  IRIS_FUSED_BN=$v timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > /dev/null 2>&1
  echo "IRIS_FUSED_BN=$v"; IRIS_FUSED_BN=$v timeout -k 10 300 python3 scripts/gpu_c4prof.py 20 2>&1 | grep "train step"
done
