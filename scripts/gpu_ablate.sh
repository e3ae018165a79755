# time the fused kernel under IRIS_ABLATE settings (diagnostic); usage: gpu_ablate.sh 0 1 3 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ab in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/abl/$ab; rm -rf $OUT; mkdir -p $OUT
  IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so IRIS_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-events --no-extras > $OUT/log 2>&1
  python3 - $OUT/b_kernel_stats.csv $ab <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_wav' in r['Name'] or 'k_minmax' in r['Name']:
        print('ablate=%s %-50s calls=%s avg_us=%.2f' % (sys.argv[2], r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
