# round 3, third GPU call: fused epilogue - suite, A/B against the two-kernel step (same binary, IRIS_EPILOGUE=1),
# rocprof stats of both, c3 engine modes after the bias_relu fix, per-step c3 / c4 statistics
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest_gpu.log
for i in 1 2; do
  for mode in 0 1; do
    IRIS_EPILOGUE=$mode timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); f=r['roofline']; print('epilogue mode $mode: ms_per_step', r['ms_per_step'], 'kernel_ms', f.get('kernel_ms'), 'median', f.get('kernel_ms_median'), 'k2', f.get('second_kernel_ms'), 'frac', f.get('frac'), 'step_frac', f['step_frac'], f.get('frac_withheld'))"
  done
done
IRIS_EPILOGUE=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-1500
for mode in 0 1; do
  IRIS_EPILOGUE=$mode timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver_e$mode -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_e$mode.json 2> $OUT/bench_driver_e$mode.err; echo "driver bench e$mode rc $?"
  grep -E "k_wav_to_mel|k_minmax" $OUT/driver_e$mode/*kernel_stats.csv | cut -c1-200
  find $OUT/driver_e$mode -name "*kernel_trace.csv" -delete
done
for mode in module engine graph; do
  timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_$mode.pre.log 2>&1; echo "c3 $mode pre rc $?"; tail -1 $OUT/c3_$mode.pre.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_$mode -o c3 -- python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_$mode.log 2>&1; echo "c3 $mode rc $?"; grep "fwd\[" $OUT/c3_$mode.log
  python3 scripts/trace_steps.py $OUT/c3_$mode/c3_kernel_trace.csv k_wav_to_mel 12 $OUT/c3_${mode}_step_kernel_stats.csv
  find $OUT/c3_$mode -name "*kernel_trace.csv" -delete
done
timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > $OUT/c4.pre.log 2>&1; echo "c4 pre rc $?"; tail -1 $OUT/c4.pre.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 scripts/gpu_c4prof.py 10 > $OUT/c4.log 2>&1; echo "c4 rc $?"; grep "train step" $OUT/c4.log
python3 scripts/trace_steps.py $OUT/c4/c4_kernel_trace.csv k_wav_to_mel 6 $OUT/c4_step_kernel_stats.csv
find $OUT/c4 -name "*kernel_trace.csv" -delete
