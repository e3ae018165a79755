# training step with / without the Winograd weight gradient, eager and as one replayed hipGraph (scripts/gpu_graph_train.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/wrw_step
echo skip-tests
for rep in 1 2; do
  for v in 0 1; do
    echo "IRIS_WINO_TRAIN_WRW=$v: "
    IRIS_WINO_TRAIN_WRW=$v timeout -k 10 200 python3 scripts/gpu_graph_train.py 30 2>&1 | grep -v amdgpu.ids | tail -4
  done
done 2>&1 | tee gpurun_out/wrw_step/c4_wrw_graph_ab.log
