# rocprofv3 kernel trace + stats of the bench command; summaries land in gpurun_out/prof
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
tail -2 $OUT/bench_stdout.log
find $OUT -name "*.csv" | head
for f in $(find $OUT -name "*kernel_stats.csv"); do cat $f; done
# keep the trace small: drop the per-dispatch trace beyond the head
for f in $(find $OUT -name "*kernel_trace.csv"); do head -50 $f > $f.head; rm $f; done
