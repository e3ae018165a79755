cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b2
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 scripts/gpu_rccl_diag.py 2>&1 | grep -v "Warning\|warn\|bucket_view\|grad.sizes\|run_backward" | tee $OUT/rccl_diag.log
bash scripts/gpu_r5_occupancy_ab.sh 2>&1 | tee $OUT/occupancy_ab.log
