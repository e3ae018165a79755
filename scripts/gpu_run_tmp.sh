cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for pr in 0 1; do
 IRIS_PAIR=$pr python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('pair $pr', 'kernel_ms', r['roofline']['kernel_ms'], 'ms_per_step', r['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
for pr in 0 1; do
 IRIS_PAIR=$pr python3 bench.py --resident --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('resident pair $pr', 'kernel_ms', r['roofline']['kernel_ms'], 'ms_per_step', r['ms_per_step'])"
done 2>&1 | tee -a $O/ab.log
PMC_OUT=pmc_pair bash scripts/gpu_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" > gpurun_out/pmc_pair.log 2>&1
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_pair/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'k_wav_to_mel' in k:
        print(k[:50], {c: round(sum(x)/len(x)/20032,1) for c,x in v.items()})
PY
