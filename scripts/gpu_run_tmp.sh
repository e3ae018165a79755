cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 600 python scripts/gpu_fuzz.py 200 7 > $O/fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 $O/fuzz.log
timeout -k 10 600 python scripts/gpu_stress.py > $O/stress.log 2>&1; echo "stress rc $?"; tail -2 $O/stress.log
