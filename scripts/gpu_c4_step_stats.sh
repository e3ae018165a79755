cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4stats
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 scripts/gpu_c4prof.py 10 2>&1 | grep "train step"
python3 scripts/trace_steps.py $OUT/c4/c4_kernel_trace.csv k_wav_to_mel 6 $OUT/c4_step_kernel_stats.csv
find $OUT/c4 -name "*kernel_trace.csv" -delete
python3 scripts/kstats.py $OUT/c4_step_kernel_stats.csv 40 6
