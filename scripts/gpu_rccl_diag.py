"""Diagnostic: per-parameter gradient differences DDP(nccl, world 1) vs plain module, in both orders, after a warm-up."""
import os, sys
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", IRIS_FORCE_PG="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from challenge_amd import sj_train as S
S.configure_miopen()
rank, world, device = S.init_distributed()
cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1', '--batch_size', '8'])
g = torch.Generator().manual_seed(100)
x = torch.randn(8, 32, 64, 1, generator=g).to(device)
y = (torch.rand(8, 2, 3, generator=g) > 0.8).float().to(device)

def fresh(ddp):
    torch.manual_seed(0)
    m = S.get_model(cfg).to(device).to(memory_format=torch.channels_last)
    m.compile(S.make_optimizer(cfg, m.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue,
              ddp=S.wrap_ddp(m, device, world) if ddp else None)
    return m

def grads(m):
    m.train()
    S.binary_crossentropy(y, m._call(x)).backward()
    torch.cuda.synchronize(device)
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

def cmp(ga, gb, tag):
    rows = []
    for n in ga:
        d = float((ga[n] - gb[n]).abs().max()) / (float(gb[n].abs().max()) + 1e-12)
        rows.append((d, n, tuple(ga[n].shape), tuple(ga[n].stride())))
    rows.sort(reverse=True)
    print(tag, "worst:", [(f"{d:.2e}", n) for d, n, _, _ in rows[:5]], flush=True)

w = grads(fresh(False))            # warm-up: MIOpen find for every shape
p1 = grads(fresh(False))
p2 = grads(fresh(False))
cmp(p2, p1, "plain vs plain")
d1 = grads(fresh(True))
cmp(d1, p1, "ddp   vs plain")
d2 = grads(fresh(True))
cmp(d2, d1, "ddp   vs ddp  ")
cmp(w, p1, "first-ever vs plain")
for var in ("IRIS_FUSED_BN",):
    pass
print("weights equal:", all(torch.equal(a, b) for a, b in zip(fresh(True).parameters(), fresh(False).parameters())))
dist.destroy_process_group()
