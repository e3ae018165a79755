# round 3, second GPU call: the GPU suite on the new code, the parity-error log, clean c3 / c4 kernel statistics
# (MIOpen find-db populated by an unprofiled run first), the driver's bench command under rocprofv3, the full bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest_gpu.log
timeout -k 10 300 python3 scripts/gpu_err_probe.py > $OUT/hip_vs_fp64.log 2>&1; echo "probe rc $?"; cat $OUT/hip_vs_fp64.log | grep -v amdgpu.ids
for mode in module engine graph; do
  timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_$mode.pre.log 2>&1; echo "c3 $mode pre rc $?"; tail -1 $OUT/c3_$mode.pre.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_$mode -o c3 -- python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_$mode.log 2>&1; echo "c3 $mode rc $?"; grep "fwd\[" $OUT/c3_$mode.log
done
timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > $OUT/c4.pre.log 2>&1; echo "c4 pre rc $?"; tail -1 $OUT/c4.pre.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 scripts/gpu_c4prof.py 10 > $OUT/c4.log 2>&1; echo "c4 rc $?"; grep "train step" $OUT/c4.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_rocprof.json 2> $OUT/bench_driver_rocprof.err; echo "driver bench rc $?"; cut -c1-300 $OUT/bench_driver_rocprof.json
for f in $(find $OUT -name "*kernel_trace.csv"); do head -300 $f > $f.head; rm $f; done
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"; cut -c1-400 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
