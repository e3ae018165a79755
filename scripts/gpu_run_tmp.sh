cd $GRAFT_REPO_ROOT
O=gpurun_out/r2u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 600 python scripts/gpu_fuzz.py 200 11 > $O/fuzz.log 2>&1; echo "fuzz rc $?"; tail -1 $O/fuzz.log
timeout -k 10 600 python scripts/gpu_stress.py > $O/stress.log 2>&1; echo "stress rc $?"; tail -1 $O/stress.log
python3 scripts/gpu_c5.py both 60 2>&1 | tail -2 | cut -c1-220
python3 scripts/gpu_shapes.py 2>&1 | grep c5
