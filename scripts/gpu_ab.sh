# A/B two builds of the library in one session: alternate runs, report bench kernel_ms (event-timed) of each
cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend.so
B=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_ab.so
for i in 1 2 3 4; do
  for L in $A $B; do
    IRIS_LIB=$L python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$(basename $L)', 'kernel_ms', r['roofline']['kernel_ms'], 'ms_per_step', r['ms_per_step'])"
  done
done
