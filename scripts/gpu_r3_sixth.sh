# round 3, sixth GPU call: GPU suite with the device-side draws, dataset rates
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/pytest_gpu.log
timeout -k 10 600 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/datasets.log
import time, torch, sys
sys.path.insert(0, '.')
from challenge_amd import sj_train as S
dev = torch.device('cuda', 0)
dcfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', '64'])
wsrc = S.synthetic_wave_sources(2, 3, 256, n_bg=16, n_voice=64, n_noise=32, seed=0)
ssrc = S.synthetic_sources(2, 3, n_bg=16, n_voice=64, n_noise=32, seed=0)
for name, fn, src in (("wave", S.make_wave_dataset, wsrc), ("spec", S.make_device_dataset, ssrc)):
    for dd in (False, True):
        it = iter(fn(dcfg, True, sources=src, device=dev, seed=0, device_draw=dd))
        for _ in range(5): next(it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): next(it)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"{name} dataset, device_draw={dd}: {1e3*dt:.3f} ms per batch of 64")
        del it
PY
