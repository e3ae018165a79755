cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b5
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "full bench rc $?"; tail -2 $OUT/bench_default.err; python3 - <<PY
import json
d = json.load(open('$OUT/bench_default.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'stft_mel_fwd_audio_s_per_s')}, d['parity']['ok'], d['roofline']['frac'], d['roofline'].get('issue_floor_us'))
e = d['extra']
print(e.get('error'))
print(json.dumps(e['c3_frontend_specaug_crnn_fwd'])[:900])
print(json.dumps(e['crnn_matrix_core_bound'])[:700])
print(json.dumps(e['c4_train_step'])[:500])
print(json.dumps(e['roofline_per_config'])[:1500])
print([ (r['batch'], r['epilogue'], r['step_us'], r['step_frac_of_8TBs']) for r in e['k1_batch_sweep']])
PY
