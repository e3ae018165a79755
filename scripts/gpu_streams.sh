cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for st in 1 2; do
  echo "=== IRIS_STREAMS=$st"
  IRIS_STREAMS=$st timeout 300 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -2
  OUT=$GRAFT_REPO_ROOT/gpurun_out/streams$st; rm -rf $OUT; mkdir -p $OUT
  IRIS_STREAMS=$st rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/log 2>&1
  grep -E "k_wav|k_minmax" $OUT/b_kernel_stats.csv | cut -c1-140
  tail -1 $OUT/log | cut -c1-160
done
