"""Does the training step with every HIP pass switched on train like the stock torch / MIOpen step?  Two models from the same
initial state, the same stream of synthetic batches (waveform-domain dataset, device-side draws replayed from the same seed),
150 steps each; prints the loss every 10 steps.  usage: python3 scripts/gpu_train_ab.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '256', '--n_chan', '1', '--batch_size', '16'])
torch.manual_seed(0)
base = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
src = S.synthetic_wave_sources(2, 3, 256, n_bg=16, n_voice=64, n_noise=32, seed=1)


def run(fused: bool):
    for name in ("FUSED_BN_RELU", "FUSED_BN_POOL", "FUSED_CONV0", "FUSED_LSTM", "FUSED_FC_BN", "WINO_TRAIN", "WINO_TRAIN_WRW", "ZERO_POOL"):
        setattr(S, name, fused)   # (round 5: the Winograd convolutions - forward, backward-data, weight gradient - and the zero pool too)
    model = copy.deepcopy(base)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    data = iter(S.make_wave_dataset(cfg, True, sources=src, device=dev, seed=7, n_fft=1024, hop=256))
    out = []
    for i in range(steps):
        loss = float(model.train_step(next(data))['loss'])
        if i % 10 == 0 or i == steps - 1:
            out.append(loss)
    return out


a, b = run(True), run(False)
a2 = run(True)
print("HIP passes, second run bit-identical to the first:", a2 == a)
print("step      " + " ".join(f"{i * 10:7d}" for i in range(len(a))))
print("HIP passes" + " ".join(f"{v:7.4f}" for v in a))
print("stock ops " + " ".join(f"{v:7.4f}" for v in b))
print(f"mean of the last 5 readings: HIP {sum(a[-5:]) / 5:.4f}   stock {sum(b[-5:]) / 5:.4f}")
