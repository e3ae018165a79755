# A/B builds of the library in one session: gpu_ab.sh name1 name2 ... (libiris_frontend_<name>.so; "prod" = the product)
# alternates runs, reports the bench's event-timed kernel_ms and ms_per_step of each
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for n in "$@"; do
    L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_$n.so
    [ "$n" = prod ] && L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend.so
    IRIS_LIB=$L python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$n', 'kernel_ms', r['roofline']['kernel_ms'], 'ms_per_step', r['ms_per_step'])"
  done
done
