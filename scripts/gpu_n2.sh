# the N > 1 control flow WITH the side measurements (DDP, no_sync A/B, InferenceEngine, bf16) on a one-GPU box: two ranks share cuda:0 over gloo
cd $GRAFT_REPO_ROOT
IRIS_BENCH_SHARE_GPU=1 timeout -k 10 800 python3 bench.py --gpus 2 --steps 20 --warmup 5 --extra-steps 5 --no-cpu-baseline > gpurun_out/n2.json 2> gpurun_out/n2.err; echo "rc $?"; tail -5 gpurun_out/n2.err; python3 - <<'PY'
import json
d = json.load(open('gpurun_out/n2.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'rccl_world', 'backend')})
print(d['ranks'])
e = d['extra']
print(json.dumps(e.get('c4_train_step'))[:600])
print(json.dumps(e.get('c3_frontend_specaug_crnn_fwd'))[:400])
print(e.get('error'))
print(d['roofline'].get('two_kernel_form'))
PY
IRIS_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --strong --no-cpu-baseline 2>/dev/null | cut -c1-700
