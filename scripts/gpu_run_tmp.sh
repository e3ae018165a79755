cd $GRAFT_REPO_ROOT
O=gpurun_out/r2y; mkdir -p $O
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic']); print(d['extra']['c3_frontend_specaug_crnn_fwd']); print(d['extra']['c4_train_step'])"
tail -3 $O/bench.err
