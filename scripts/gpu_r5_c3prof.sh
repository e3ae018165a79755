cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_c3prof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 engine > $OUT/c3_engine.pre.log 2>&1; tail -1 $OUT/c3_engine.pre.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_engine -o c3 -- python3 scripts/gpu_fwdprof.py 20 engine > $OUT/c3_engine.log 2>&1; grep "fwd\[" $OUT/c3_engine.log
python3 scripts/trace_steps.py $OUT/c3_engine/c3_kernel_trace.csv k_wav_to_mel 12 $OUT/c3_engine_step_kernel_stats.csv
find $OUT/c3_engine -name "*kernel_trace.csv" -delete
python3 scripts/kstats.py $OUT/c3_engine_step_kernel_stats.csv 25 12
