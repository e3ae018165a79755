"""c4 training step (fused frontend with SpecAugment + CRNN v9 forward, loss, backward, AGC + clipvalue, Adam; batch 64,
1 GPU) under rocprofv3 --kernel-trace --stats: which kernels the step is made of.
usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/gpu_c4prof.py [n] [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
batch, length = 64, 130816
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, batch, length, dev, training=True, device_draw=True, seed=99)
gen = torch.Generator(device=dev).manual_seed(4321)
wav = torch.randn(batch, 1, length, generator=gen, device=dev) * 0.1
y = (torch.rand(batch, 16, 3, generator=gen, device=dev) < 0.1).float()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"


def step():
    if bf16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            model.train_step((fe(wav), y))
    else:
        model.train_step((fe(wav), y))


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"train step{' (bf16 autocast)' if bf16 else ''} {1e3 * (time.perf_counter() - t0) / n:.3f} ms per batch of {batch}")
