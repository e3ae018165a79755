cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r2a/pytest_gpu3.log
timeout 300 python scripts/gpu_c5.py both 2>&1 | tee gpurun_out/r2a/c5.log
timeout 300 python scripts/gpu_dataset.py 2>&1 | tail -5 | tee gpurun_out/r2a/dataset.log
