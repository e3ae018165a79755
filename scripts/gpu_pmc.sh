# PMC counter passes (no tracing flags besides --kernel-trace), one pass per counter group.
# usage: gpu_pmc.sh "<counters pass 1>" "<counters pass 2>" ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-extras > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r.get('Kernel_Name','')[:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'k_wav' in k or 'k_minmax' in k:
        print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
PY
done
