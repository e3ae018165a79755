cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r2a/pytest_gpu2.log
