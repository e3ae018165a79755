cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_hp.so timeout -k 10 600 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -3
bash scripts/gpu_ab.sh prod hp 2>&1 | tee $O/ab.log
BENCH_ARGS="--resident" bash scripts/gpu_ab.sh prod hp 2>&1 | tee $O/ab_res.log
for v in prod hp; do L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_$v.so; [ $v = prod ] && L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend.so
 IRIS_LIB=$L python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --only-sweep 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$v', [ (x['batch'], x['k1_us'], x['k1_frac_of_8TBs']) for x in r.get('extra',{}).get('k1_batch_sweep',[])])"
done 2>&1 | tee $O/sweep.log
