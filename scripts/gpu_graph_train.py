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
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=torch.tensor(cfg.lr, device=dev), eps=1e-7, fused=True, capturable=True)
model.compile(opt, S.binary_crossentropy, clipvalue=cfg.clipvalue)
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

# capture: the documented whole-network pattern (grads dropped inside the capture: they live in the graph's pool)
x_static = fe(wav).clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        model.train_step((x_static, y))
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
object.__setattr__(model, '_fused_agc', None)
model.optimizer.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    model.train()
    y_pred = model._call(x_static)
    loss = model.loss_fn(y, y_pred)
    loss.backward()
    if model._fused_agc is None:
        object.__setattr__(model, '_fused_agc', S.FusedAGC(list(model.parameters())))
    model._fused_agc(0.01, 1e-3, model.clipvalue)
    model.optimizer.step()
    model.optimizer.zero_grad(set_to_none=True)
torch.cuda.synchronize()


def graphed():
    x_static.copy_(fe(wav))
    g.replay()
    return loss


l0 = float(graphed())
for _ in range(5):
    graphed()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    graphed()
torch.cuda.synchronize()
print(f"graphed train step {1e3 * (time.perf_counter() - t0) / n:.3f} ms   loss {l0:.4f} -> {float(loss):.4f}")
