cd $GRAFT_REPO_ROOT
for i in 1 2; do for ev in "--event-every 4" "--event-every 10" "--event-every 20" "--no-kernel-events"; do
 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extras $ev 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$ev', 'ms_per_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'], 'n', r['roofline']['launches_timed'])"
done; done
