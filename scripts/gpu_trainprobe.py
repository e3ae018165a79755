import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from challenge_amd import sj_train as S
dev = torch.device("cuda", 0)
batch, length = 64, 130816
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
cl = os.environ.get("CL", "1") == "1"
model = S.get_model(cfg).to(dev)
if cl: model = model.to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, batch, length, dev, training=True)
wav = torch.randn(batch, 1, length, device=dev) * 0.1
y = (torch.rand(batch, 16, 3, device=dev) < 0.1).float()
def sync(): torch.cuda.synchronize()
for i in range(8):
    sync(); t0 = time.perf_counter()
    x = fe(wav); sync(); t1 = time.perf_counter()
    model.train(); model.optimizer.zero_grad(set_to_none=True)
    yp = model(x); loss = S.binary_crossentropy(y, yp); sync(); t2 = time.perf_counter()
    loss.backward(); sync(); t3 = time.perf_counter()
    params = [p for p in model.parameters() if p.grad is not None]
    new = S.adaptive_clip_grad(params, [p.grad for p in params])
    for p, g in zip(params, new): p.grad = g
    torch.nn.utils.clip_grad_value_(params, cfg.clipvalue); sync(); t4 = time.perf_counter()
    model.optimizer.step(); sync(); t5 = time.perf_counter()
    print(f"iter {i}: frontend {1e3*(t1-t0):.2f} fwd {1e3*(t2-t1):.2f} bwd {1e3*(t3-t2):.2f} agc+clip {1e3*(t4-t3):.2f} adam {1e3*(t5-t4):.2f} ms", flush=True)
