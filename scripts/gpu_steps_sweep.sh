cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 -m pytest tests/test_frontend_gpu.py -x -q -m gpu -k "prepared or captured" 2>&1 | tail -2
for i in 1 2 3; do
for k in 20 50 200 1000; do
  timeout -k 10 200 python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('steps $k: ms_per_step', r['ms_per_step'])"
done
done
