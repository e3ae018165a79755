cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_driver
rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench_driver_cmd_$i.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver$i -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd_rocprof_$i.json 2> $OUT/rocprof_$i.err
grep -E "k_wav_to_mel<10, 0, false, false, 1, 1>" $OUT/driver$i/*kernel_stats.csv | cut -c1-120
find $OUT/driver$i -name "*kernel_trace.csv" -delete
python3 - <<PY
import json
d = json.load(open('$OUT/bench_driver_cmd_$i.json')); r = d['roofline']
print('plain $i: value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'], 'parity', d['parity']['ok'], 'cpu', d['cpu_baseline']['value'])
d = json.loads([l for l in open('$OUT/bench_driver_cmd_rocprof_$i.json') if l.startswith('{')][-1]); r = d['roofline']
print('under rocprofv3 $i: ms/step', d['ms_per_step'], 'kernel_ms (events)', r['kernel_ms'])
PY
done
