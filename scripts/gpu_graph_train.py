"""Whole training step (forward, loss, backward, AGC + clipvalue, Adam) as ONE replayed hipGraph against the eager step.
usage: python3 scripts/gpu_graph_train.py [n]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def eager():
    return model.train_step((fe(wav), y))['loss']


for _ in range(5):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    eager()
torch.cuda.synchronize()
print(f"eager   train step {1e3 * (time.perf_counter() - t0) / n:.3f} ms")

# the same step captured once and replayed (sj_train.GraphedTrainStep, what bench.py times as c4_train_step.hipgraph)
gm = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
gm.load_state_dict(model.state_dict())
gm.compile(S.make_optimizer(cfg, gm.parameters(), capturable=True), S.binary_crossentropy, clipvalue=cfg.clipvalue)
gstep = S.GraphedTrainStep(gm, (fe(wav), y))
for _ in range(5):
    out = gstep((fe(wav), y))
torch.cuda.synchronize()
l0 = float(out['loss'])
t0 = time.perf_counter()
for _ in range(n):
    out = gstep((fe(wav), y))
torch.cuda.synchronize()
print(f"graphed train step {1e3 * (time.perf_counter() - t0) / n:.3f} ms   loss {l0:.4f} -> {float(out['loss']):.4f}")
