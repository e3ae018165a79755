# rocprofv3 kernel stats of the on-device dataset at the reference's default shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/dsprof; rm -rf $OUT; mkdir -p $OUT
IRIS_DATASET_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ds -- python3 scripts/gpu_dataset.py > $OUT/log 2>&1
tail -2 $OUT/log
head -12 $OUT/ds_kernel_stats.csv | cut -c1-150
