# seeded accuracy sweep of the fused kernel against the fp64 oracle -> gpurun_out/hip_vs_fp64_sweep.log
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# keep freed arrays in the heap: fresh pages are slow to fault in on these boxes
export MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=100000000000
timeout -k 10 ${SWEEP_TIMEOUT:-1000} python scripts/gpu_err_sweep.py "$@" > gpurun_out/hip_vs_fp64_sweep.log 2> gpurun_out/hip_vs_fp64_sweep.err
rc=$?
tail -30 gpurun_out/hip_vs_fp64_sweep.log; tail -5 gpurun_out/hip_vs_fp64_sweep.err
exit $rc
