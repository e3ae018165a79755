# Round-5 K1 occupancy A/B: the SAME instruction stream (band weights and window from LDS: 88 VGPRs) at
# 16 waves per CU (b16: one workgroup), 20 (c20: two workgroups of 10 = five waves per SIMD) and 16 as two workgroups (d16),
# beside the product kernel (register constants, 127 VGPRs, 16 waves) - two-kernel form (no epilogue in K1), c2, HBM-rotating
cd $GRAFT_REPO_ROOT
export IRIS_EPILOGUE=1
L=$GRAFT_REPO_ROOT/challenge_amd/csrc
IRIS_LIB=$L/libiris_frontend_c20.so timeout -k 10 600 python3 -m pytest tests/test_frontend_gpu.py -x -q -k "golden or c2_full or bands_all" 2>&1 | tail -2
for i in 1 2 3; do
  for n in prod b16 c20 d16; do
    LIB=$L/libiris_frontend_$n.so; [ "$n" = prod ] && LIB=$L/libiris_frontend.so
    IRIS_LIB=$LIB timeout -k 10 300 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); ro=r['roofline']; print('$n', ro['kernel'], 'k1_us', round(1e3*ro['kernel_ms'],2), 'median', round(1e3*ro['kernel_ms_median'],2), 'k2_us', round(1e3*(ro['second_kernel_ms'] or 0),2), 'step_us', round(1e3*r['ms_per_step'],2))"
  done
done
# large batch: the frame loop dominates (B = 512)
for n in prod b16 c20 d16; do
  LIB=$L/libiris_frontend_$n.so; [ "$n" = prod ] && LIB=$L/libiris_frontend.so
  IRIS_LIB=$LIB timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-sweep --no-cpu-baseline --no-kernel-events 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$n', ' | '.join(f\"B {x['batch']}: k1 {x['k1_us']} step {x['step_us']}\" for x in r['extra']['k1_batch_sweep']))"
done
