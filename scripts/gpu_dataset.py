"""Input-pipeline rates at the reference's default shape: per-sample drop-in graph vs the batched
on-device dataset (both produce (log-mel [B, 80, 512, 2], labels [B, 16, 3]))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from challenge_amd import sj_train as S

dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', '64'])
sources = S.synthetic_sources(2, 3, freq=257, n_bg=16, n_voice=64, n_noise=32, seed=0)

def rate(ds, n, warm=2):
    it = iter(ds)
    for _ in range(warm): next(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): x, y = next(it)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    return dt, x.shape, y.shape

dt, xs, ys = rate(S.make_device_dataset(cfg, True, sources=sources, device=dev, seed=0), 30)
audio_s = 64 * 512 * 256 / 16000
print(f"device dataset : {dt*1e3:8.2f} ms/batch  {audio_s/dt:10.0f} audio-s/s  x{tuple(xs)} y{tuple(ys)}", flush=True)
if os.environ.get('IRIS_DATASET_ONLY'):
    sys.exit(0)
dt, xs, ys = rate(S.make_dataset(cfg, True, sources=sources), 3, warm=1)
print(f"per-sample graph: {dt*1e3:8.2f} ms/batch  {audio_s/dt:10.0f} audio-s/s  x{tuple(xs)} y{tuple(ys)}", flush=True)
