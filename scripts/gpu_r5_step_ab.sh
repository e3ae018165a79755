# the training step eager / graphed with one switch flipped: usage  bash scripts/gpu_r5_step_ab.sh ENVVAR [reps]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/step_ab
VAR=${1:-IRIS_ZERO_POOL}
for rep in $(seq 1 ${2:-2}); do
  for v in 0 1; do
    echo "$VAR=$v: "
    env $VAR=$v timeout -k 10 200 python3 scripts/gpu_graph_train.py 30 2>&1 | grep -v amdgpu.ids | tail -2
  done
done 2>&1 | tee gpurun_out/step_ab/$VAR.log
