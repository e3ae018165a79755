"""FUSED_PACK on / off: the same eager train steps from the same state must give the same bits.  Prints the first difference."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
from challenge_amd import hip_autograd as HA
S.configure_miopen()
dev = torch.device("cuda", 0)
B, T = 16, 256
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', str(T), '--n_chan', '1', '--batch_size', str(B)])
torch.manual_seed(0)
base = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
x = torch.rand(B, 64, T, 1, device=dev)
y = (torch.rand(B, T // 32, 3, device=dev) < 0.1).float()


def run(fused, steps=4):
    S.FUSED_PACK = fused
    HA._PACKS.clear()
    m = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    m.load_state_dict(base.state_dict())
    m.compile(S.make_optimizer(cfg, m.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    hist = []
    for s in range(steps):
        loss = m.train_step((x, y))['loss']
        hist.append((float(loss), [p.grad.detach().clone() for p in m.parameters()], [p.detach().clone() for p in m.parameters()]))
    return m, hist


ma, ha = run(True)
mb, hb = run(False)
names = [n for n, _ in ma.named_parameters()]
for s, (a, b) in enumerate(zip(ha, hb)):
    gd = [(float((u - v).abs().max()) / (float(v.abs().max()) + 1e-30), n) for n, u, v in zip(names, a[1], b[1]) if not torch.equal(u, v)]
    pd = [n for n, u, v in zip(names, a[2], b[2]) if not torch.equal(u, v)]
    print(f"step {s}: loss {a[0]:.7f} / {b[0]:.7f}; gradients differing: {len(gd)} worst {max(gd) if gd else None}; parameters differing: {len(pd)} {pd[:4]}")


def run_graph(fused, steps=6, size=(16, 256)):
    S.FUSED_PACK = fused
    HA._PACKS.clear()
    b, t = size
    c = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', str(t), '--n_chan', '1', '--batch_size', str(b)])
    torch.manual_seed(0)
    src = S.get_model(c).to(dev).to(memory_format=torch.channels_last)
    m = S.get_model(c).to(dev).to(memory_format=torch.channels_last)
    m.load_state_dict(src.state_dict())
    m.compile(S.make_optimizer(c, m.parameters(), capturable=True), S.binary_crossentropy, clipvalue=c.clipvalue)
    g = torch.Generator(device=dev).manual_seed(5)
    xs = [torch.rand(b, 64, t, 1, device=dev, generator=g) for _ in range(steps)]
    ys = [(torch.rand(b, t // 32, 3, device=dev, generator=g) < 0.1).float() for _ in range(steps)]
    gs = S.GraphedTrainStep(m, (xs[0], ys[0]), preserve_state=True)
    hist = []
    for s in range(steps):
        loss = gs((xs[s], ys[s]))['loss']
        torch.cuda.synchronize()
        hist.append((float(loss), [p.detach().clone() for p in m.parameters()]))
    # and the same batches eagerly, from the same start (the reference trajectory)
    e = S.get_model(c).to(dev).to(memory_format=torch.channels_last)
    e.load_state_dict(src.state_dict())
    e.compile(S.make_optimizer(c, e.parameters()), S.binary_crossentropy, clipvalue=c.clipvalue)
    eh = []
    for s in range(steps):
        loss = e.train_step((xs[s], ys[s]))['loss']
        eh.append((float(loss), [p.detach().clone() for p in e.parameters()]))
    return hist, eh


for size in ((16, 256), (64, 512)):
    ga, ea = run_graph(True, size=size)
    gb, eb = run_graph(False, size=size)
    for s in range(len(ga)):
        def worst(u, v):
            return max(float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30) for a, b in zip(u, v))
        print(f"{size} step {s}: graph fused vs graph per-layer {worst(ga[s][1], gb[s][1]):.1e}; graph fused vs eager fused {worst(ga[s][1], ea[s][1]):.1e}; "
              f"graph per-layer vs eager per-layer {worst(gb[s][1], eb[s][1]):.1e}; eager fused vs eager per-layer {worst(ea[s][1], eb[s][1]):.1e}; "
              f"loss {ga[s][0]:.6f} {gb[s][0]:.6f} {ea[s][0]:.6f} {eb[s][0]:.6f}")
