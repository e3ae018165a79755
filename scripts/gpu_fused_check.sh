# GPU suite + where the fused kernel's time goes (diag stamps) + the bench at 200 and 20 steps, twice
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${CHECK_OUT:-fused_check}
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so IRIS_ABLATE=512 timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-events 2>&1 | grep "iris dbg" | tee -a $OUT/epilogue_phases.log
for i in 1 2; do
  for k in 200 20; do
    timeout -k 10 200 python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); f=r['roofline']; print('steps $k: ms_per_step', r['ms_per_step'], 'kernel_ms', f.get('kernel_ms'), 'median', f.get('kernel_ms_median'), 'frac', f.get('frac'), 'step_frac', f['step_frac'], f.get('frac_withheld'))"
  done
done | tee $OUT/bench_ab.log
