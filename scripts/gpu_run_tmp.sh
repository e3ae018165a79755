cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout -k 10 120 ./scripts/microbench/fft_pair 2>&1 | tee gpurun_out/r3g/fft_pair.log
