# round 5, first GPU batch: the whole GPU suite, the driver command with its parity field, RCCL at world size 1 (test, bench
# line, kernel trace), a 4-rank control-flow rehearsal sharing the GPU (gloo; the pool's process guard allows at most 6)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b1
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
grep -E "rccl world 1|rule ratio" $OUT/pytest_gpu.log | cut -c1-400 | tail -12
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc $?"; python3 -c "
import json; d = json.load(open('$OUT/bench_driver_cmd.json')); print({k: d[k] for k in ('value', 'ms_per_step', 'parity')}); print(d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
IRIS_FORCE_PG=1 timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra-steps 10 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err; echo "rccl bench rc $?"; tail -3 $OUT/bench_rccl_world1.err; python3 -c "
import json; d = json.load(open('$OUT/bench_rccl_world1.json')); print({k: d.get(k) for k in ('value', 'backend', 'backend_is_rccl', 'rccl_world', 'train_step_ms', 'allreduce_exposed_ms', 'grad_bytes', 'forced_process_group')}); print(d['extra']['c4_train_step']['allreduce'], d['extra'].get('error'))"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rccl_w1 -o rccl_w1 -- python3 scripts/gpu_rccl_world1.py > $OUT/rccl_w1.log 2>&1; echo "rocprof rc $?"; grep "^{" $OUT/rccl_w1.log
grep -i -E "nccl|rccl" $OUT/rccl_w1/*kernel_stats.csv | cut -c1-220
python3 scripts/kstats.py $(ls $OUT/rccl_w1/*kernel_stats.csv | head -1) 12 > $OUT/rccl_w1_top.txt 2>&1; head -16 $OUT/rccl_w1_top.txt
grep -i -E "nccl|rccl" $OUT/rccl_w1/*kernel_trace.csv | head -40 > $OUT/rccl_w1_nccl_dispatches.csv
find $OUT/rccl_w1 -name "*kernel_trace.csv" -delete
IRIS_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_share4.json 2> $OUT/bench_share4.err; echo "share4 rc $?"; python3 -c "
import json; d = json.load(open('$OUT/bench_share4.json')); print(d['n_gpus'], d['rccl_world'], d['backend'], len(d['ranks']), [r['rank'] for r in d['ranks']])"
bash scripts/gpu_r5_sweep_ab.sh 2>&1 | tee $OUT/sweep_ab.log
