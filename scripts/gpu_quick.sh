# frontend GPU tests + rocprof kernel stats of the bench + phase breakdown
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -5
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/log 2>&1
tail -1 $OUT/log | cut -c1-400
python3 - $OUT/b_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_' in r['Name']:
        print('%-60s calls=%s avg_us=%.2f min_us=%.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
bash scripts/gpu_phase.sh 4608 2>&1 | grep "iris dbg"
