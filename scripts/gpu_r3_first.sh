# round 3, first GPU call: parity-error log, c3 / c4 kernel stats, the driver's bench command as it stands
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
rocminfo | grep -E "Marketing Name|gfx" | head -2
timeout -k 10 300 python3 scripts/gpu_err_probe.py > $OUT/hip_vs_fp64.log 2>&1; echo "probe rc $?"; cat $OUT/hip_vs_fp64.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 scripts/gpu_fwdprof.py 20 > $OUT/c3.log 2>&1; echo "c3 rc $?"; tail -2 $OUT/c3.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 scripts/gpu_c4prof.py 10 > $OUT/c4.log 2>&1; echo "c4 rc $?"; tail -2 $OUT/c4.log
for f in $(find $OUT -name "*kernel_trace.csv"); do head -400 $f > $f.head; rm $f; done
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"; cut -c1-600 $OUT/bench_driver.json
find $OUT -name "*kernel_stats.csv"
