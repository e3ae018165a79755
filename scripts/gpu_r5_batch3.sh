cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b3
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_ddp_gpu.py -x -q -s 2>&1 | grep -v "Warning\|warn\|bucket_view\|grad.sizes\|run_backward\|amdgpu.ids" | tail -15 | tee $OUT/pytest_ddp.log
