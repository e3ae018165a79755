# PMC counter passes (no tracing flags besides --kernel-trace), ONE pass per counter group: derived TCC counters
# (FETCH_SIZE, WRITE_SIZE) each need a pass of their own ("exceeds the capabilities of the hardware" otherwise, and
# the crashed tool then hangs - hence the timeout around every pass).
# usage: gpu_pmc.sh "<counters pass 1>" "<counters pass 2>" ...   then, in the build container:
#        python scripts/pmc_summarise.py   (-> profiles/r2/pmc_traffic.json)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PMC_OUT:-pmc}
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "$@"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 ${PMC_CMD:-bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-extras $BENCH_ARGS} > $OUT/p$i.log 2>&1
  rc=$?; echo "  rc $rc"; tail -1 $OUT/p$i.log | cut -c1-200
  if [ $rc -ge 124 ]; then echo "pass $i was killed at its limit: no further pass"; exit $rc; fi   # no GPU step after a hang
  find $OUT/p$i -name "*kernel_trace.csv" -delete
done
find $OUT -name "*counter_collection.csv" | head -30; du -sh $OUT
