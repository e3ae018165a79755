# round 3, fifth GPU call: host wake-up latency of the closing synchronize (spin vs blocking), cross-lane issue rates
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  for blk in 0 1; do
    for k in 20 200; do
      IRIS_BENCH_BLOCKING_SYNC=$blk timeout -k 10 200 python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('blocking_sync=$blk steps $k: ms_per_step', r['ms_per_step'])"
    done
  done
done | tee $OUT/fence_ab.log
timeout -k 10 120 scripts/microbench/xlane_rate | tee $OUT/microbench_xlane_rate.log
