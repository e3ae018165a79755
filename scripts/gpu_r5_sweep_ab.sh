# batch sweep of the c2 step (B = 32, 128, 256, 512) in the three forms of the epilogue, same session:
# default (LDS tile where it fits, in place beyond), IRIS_EPILOGUE=1 (two kernels), IRIS_EPILOGUE=2 (in place everywhere)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_sweep
mkdir -p $OUT
for rep in 1 2; do
for mode in default 1 2; do
  if [ $mode = default ]; then unset IRIS_EPILOGUE; else export IRIS_EPILOGUE=$mode; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-sweep --no-cpu-baseline --no-kernel-events > $OUT/sweep_${mode}_$rep.json 2> $OUT/sweep_${mode}_$rep.err || { echo "sweep $mode failed"; tail -3 $OUT/sweep_${mode}_$rep.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open('$OUT/sweep_${mode}_$rep.json'))
print('epilogue $mode rep $rep: c2 step', round(1e3 * d['ms_per_step'], 2), 'us |', ' | '.join(f"B {r['batch']}: {r['epilogue']} k1 {r['k1_us']} k2 {r['second_kernel_us']} step {r['step_us']} us = {r['step_frac_of_8TBs']}" for r in d['extra']['k1_batch_sweep']))
PY
done
done
