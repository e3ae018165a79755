cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c3prof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_transforms_gpu.py -x -q -m gpu -k "bias_relu or inference_engine" 2>&1 | tail -2
timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 engine > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e -o c3 -- python3 scripts/gpu_fwdprof.py 20 engine 2>&1 | grep "fwd\["
python3 scripts/trace_steps.py $OUT/e/c3_kernel_trace.csv k_wav_to_mel 12 $OUT/c3_engine_step_kernel_stats.csv
python3 scripts/kstats.py $OUT/c3_engine_step_kernel_stats.csv 14 12
find $OUT -name "*kernel_trace.csv" -delete
