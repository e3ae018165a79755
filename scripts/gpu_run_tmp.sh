cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
bash scripts/gpu_ab.sh prod nox noxu 2>&1 | tee $O/ab.log
BENCH_ARGS="--resident" bash scripts/gpu_ab.sh prod nox noxu 2>&1 | tee $O/ab_res.log
