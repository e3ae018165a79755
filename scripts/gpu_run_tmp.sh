cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do for st in 0 1; do
 IRIS_STEAL=$st python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('steal $st', 'kernel_ms', r['roofline']['kernel_ms'], 'ms_per_step', r['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
for st in 0 1; do
 IRIS_STEAL=$st python3 bench.py --resident --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('resident steal $st', 'kernel_ms', r['roofline']['kernel_ms'], 'ms_per_step', r['ms_per_step'])"
done 2>&1 | tee -a $O/ab.log
