# what a 20-step timed window looks like on the GPU: per-dispatch start / end of the fused kernel around the window
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/window
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o w -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-events > $OUT/bench.json 2>/dev/null
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/window/t/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if 'k_wav_to_mel' in r['Kernel_Name']]
rows.sort()
# the timed window = the last 20 dispatches; the 5 warm-up ones and the tail of the pre-conditioning burst sit in front
tail = rows[-60:]
prev = None
for i, (s, e) in enumerate(tail):
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{i - 40:4d} dur {(e - s) / 1e3:7.2f} us  gap before {gap:8.2f} us")
    prev = e
w = rows[-20:]
print("window: first start -> last end", (w[-1][1] - w[0][0]) / 1e3, "us; sum of durations", sum(e - s for s, e in w) / 1e3)
PY
cut -c1-200 $OUT/bench.json
