cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
bash scripts/gpu_ab.sh prod sw 2>&1 | tee $O/ab.log
BENCH_ARGS="--resident" bash scripts/gpu_ab.sh prod sw 2>&1 | tee $O/ab_res.log
