cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b6
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_transforms_gpu.py -x -q -k "winograd_training" 2>&1 | tail -2
for rep in 1 2 3; do
for cfg in "0 128 64" "1 128 64" "1 64 64" "1 128 128"; do set -- $cfg
IRIS_WINO_TRAIN=$1 IRIS_WINO_TRAIN_MIN_C_FWD=$2 IRIS_WINO_TRAIN_MIN_C_BWD=$3 timeout -k 10 300 python3 scripts/gpu_c4prof.py 20 2>&1 | grep "train step" | sed "s/^/IRIS_WINO_TRAIN=$1 FWD>=$2 BWD>=$3: /" | tee -a $OUT/c4_wino_train_ab.log
done
done
