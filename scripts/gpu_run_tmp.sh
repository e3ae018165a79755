cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python scripts/gpu_fuzz.py 700 2026 > $O/fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 $O/fuzz.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source')); print(d['extra']['device_dataset']['ms_per_batch'], d['extra']['wave_dataset'])"
