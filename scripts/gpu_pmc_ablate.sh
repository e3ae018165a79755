cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ab in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcabl/$ab; rm -rf $OUT; mkdir -p $OUT
  IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so IRIS_ABLATE=$ab rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extras > $OUT/log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$ab" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_wav' in r.get('Kernel_Name',''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
fr=20032.0
print('ablate=%s'%sys.argv[2], {c.replace('SQ_',''): round(sum(x)/len(x)/fr,1) for c,x in sorted(agg.items())})
PY
done
