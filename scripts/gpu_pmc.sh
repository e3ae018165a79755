# PMC counter passes (no tracing flags besides --kernel-trace), one pass per counter group.
# usage: gpu_pmc.sh "<counters pass 1>" "<counters pass 2>" ...   then, in the build container:
#        python scripts/pmc_summarise.py   (-> profiles/r2/pmc_traffic.json)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-extras $BENCH_ARGS > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log | cut -c1-300
done
find $OUT -name "*counter_collection.csv" | head; du -sh $OUT
