cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
python3 - <<'PY'
# input-pipeline rate of the waveform-domain dataset at the reference's default shape
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from challenge_amd import sj_train as S
dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', '64'])
ds = iter(S.make_wave_dataset(cfg, True, sources=S.synthetic_wave_sources(2, 3, 256, n_bg=16, n_voice=64, n_noise=32, seed=0), device=dev, seed=0))
for _ in range(5): next(ds)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): next(ds)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"make_wave_dataset: {1e3*dt:.3f} ms per batch of 64 = {64*512*256/16000/dt/1e6:.2f} M audio-s/s")
PY
