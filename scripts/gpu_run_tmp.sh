cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -5
bash scripts/gpu_ab.sh prod notile 2>&1 | tee $O/ab.log
BENCH_ARGS="--resident" bash scripts/gpu_ab.sh prod notile 2>&1 | tee $O/ab_res.log
