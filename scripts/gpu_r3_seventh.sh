# round 3, seventh GPU call: STFT write-out A/B (shapes), spec-domain device-draw dataset profile
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_frontend_gpu.py -x -q -m gpu -k "stft or impulse or minimal" 2>&1 | tail -3
timeout -k 10 300 python3 scripts/gpu_shapes.py 2>&1 | grep -v amdgpu.ids | tee $OUT/shapes.log
cat > /tmp/dsprof.py <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
from challenge_amd import sj_train as S
dev = torch.device('cuda', 0)
dd = sys.argv[1] == '1'
dcfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', '64'])
ssrc = S.synthetic_sources(2, 3, n_bg=16, n_voice=64, n_noise=32, seed=0)
it = iter(S.make_device_dataset(dcfg, True, sources=ssrc, device=dev, seed=0, device_draw=dd))
for _ in range(5): next(it)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): next(it)
torch.cuda.synchronize(); print("device_draw", dd, (time.perf_counter() - t0) / 30 * 1e3, "ms per batch")
PY
for dd in 0 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ds$dd -o ds -- python3 /tmp/dsprof.py $dd 2>&1 | grep "device_draw"
  python3 scripts/kstats.py $OUT/ds$dd/ds_kernel_stats.csv 8 35
  find $OUT/ds$dd -name "*kernel_trace.csv" -delete
done
