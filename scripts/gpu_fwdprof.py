"""c3 forward (frontend + SpecAugment + CRNN v9, batch 64) under rocprofv3 --kernel-trace --stats:
which kernels the 6 ms are made of.   usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/gpu_fwdprof.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
batch, length = 64, 130816
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last).eval()
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, batch, length, dev, training=True, device_draw=True, seed=99)
wav = torch.randn(batch, 1, length, device=dev) * 0.1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(3):
        model(fe(wav))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(fe(wav))
    torch.cuda.synchronize()
print(f"fwd {1e3 * (time.perf_counter() - t0) / n:.3f} ms per batch of {batch}")
