"""Where do the ~88 small elementwise launches of a training step come from, and do they go away when the gradients are
dropped (zero_grad(set_to_none=True)) so that AccumulateGrad can take the incoming gradient instead of adding it?"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', '64'])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
x = torch.rand(64, 64, 512, 1, device=dev)
y = (torch.rand(64, 16, 3, device=dev) < 0.1).float()
for _ in range(3):
    model.train_step((x, y))
torch.cuda.synchronize()

def profile(fn, tag):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    c = collections.Counter()
    t = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            name = e.name.split('<')[0].split('(')[0][:60]
            c[name] += 1
            t[name] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    tot = sum(t.values()) / 3e3
    print(f"--- {tag}: {sum(c.values()) // 3} launches / step, {tot:.3f} ms device time / step")
    for k, v in t.most_common(14):
        print(f"   {c[k] // 3:4d} x {k:60s} {v / 3e3:.3f} ms")

profile(lambda: model.train_step((x, y)), "grads kept in place (product)")

def step_none():
    model.train()
    model.optimizer.zero_grad(set_to_none=True)
    loss = model.loss_fn(y, model(x))
    loss.backward()
    if model._fused_agc is None:
        object.__setattr__(model, '_fused_agc', S.FusedAGC(list(model.parameters())))
    model._fused_agc(0.01, 1e-3, model.clipvalue)
    model.optimizer.step()
for _ in range(8):
    step_none()
sigs = set()
for _ in range(8):
    step_none()
    sigs.add(tuple(p.grad.data_ptr() for p in model.parameters()))
print("distinct gradient address sets over 8 steps:", len(sigs))
profile(step_none, "gradients dropped every step (set_to_none)")
import time
for fn, tag in ((lambda: model.train_step((x, y)), "kept"), (step_none, "dropped")):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); print(tag, f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms / step")
