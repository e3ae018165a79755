"""Hunt for the tail of the c4 gradient-parity figure (one 8.2e-5 at features.0.convs.1.0.weight in ~25 evaluations, everything
else 4e-6 .. 2e-5; the step was bit-reproducible and its forward 2e-7 from the reference in that run).

Per trial (fresh weights state every few trials, fresh waveforms, labels and SpecAugment bands every trial), at batch 64 x 512 frames:
product forward / backward (every HIP pass on) -> Decisions -> fp64 RefCRNN with those decisions; worst gradient error and where.
Whenever a trial reads above `--flag` (default 2.5e-5) it ALSO runs the STOCK fp32 layers with the same forced decisions and prints,
for the worst tensor: the stock fp32 error on it, how the deviation is spread over the output channels (one wrong element of dz
touches one output channel's row of dW), and the three largest deviations - enough to tell fp32 conditioning of that batch (stock
reads the same) from a wrong element (stock reads 1e-5, deviation in one row).
usage: python3 scripts/gpu_c4_parity_hunt.py [trials] [--split] [--flag 2.5e-5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
from challenge_amd.hip_autograd import record_activations
from oracle import crnn_parity as P, crnn_ref as R

args = [a for a in sys.argv[1:] if not a.startswith("--")]
trials = int(args[0]) if args else 30
flag = float(sys.argv[sys.argv.index("--flag") + 1]) if "--flag" in sys.argv else 2.5e-5
S.WINO_SPLIT_BF16 = "--split" in sys.argv
S.configure_miopen()
dev = torch.device("cuda", 0)
B, T = 64, 512
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', str(T), '--n_chan', '1', '--batch_size', str(B)])
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, B, (T - 1) * 256, dev, training=True, device_draw=True, seed=7)
gen = torch.Generator(device=dev).manual_seed(2024)


def rel_rows(mine, theirs, names):
    return sorted(((P._rel(a, b), n, k) for k, (n, a, b) in enumerate(zip(names, mine, theirs)) if not P._bn_fed_bias(n)), reverse=True)


model = None
hist = []
t_start = time.time()
for trial in range(trials):
    if trial % 5 == 0:   # a fresh model, a few Adam steps into training (what the bench's leg sees)
        torch.manual_seed(100 + trial)
        model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
        model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
        wav0 = torch.randn(B, 1, (T - 1) * 256, generator=gen, device=dev) * 0.1
        y0 = (torch.rand(B, T // 32, 3, generator=gen, device=dev) < 0.1).float()
        for _ in range(3 * (trial // 5 % 4)):
            model.train_step((fe(wav0), y0))
    wav = torch.randn(B, 1, (T - 1) * 256, generator=gen, device=dev) * (0.03 + 0.1 * float(torch.rand((), generator=gen, device=dev)))
    y = (torch.rand(B, T // 32, 3, generator=gen, device=dev) < 0.1).float()
    feats = fe(wav)
    names = [n for n, _ in model.named_parameters()]
    m = P.clone_module(model).train()
    for p in m.parameters():
        p.grad = None
    td = {}
    hook = m.td.register_forward_hook(lambda mod, i, o: td.__setitem__('z', o.detach()))
    with record_activations() as acts:
        out = m(feats)
    hook.remove()
    S.binary_crossentropy(y, out).backward()
    torch.cuda.synchronize()
    g = [p.grad.detach().clone() for p in m.parameters()]
    d = R.Decisions(acts, td['z'])
    del acts
    r64 = R.RefCRNN(64, T, 1, 9).to(dev).double().load_from(model)
    q = R.reference_step(r64, feats, y, decisions=d)
    rows = rel_rows(g, q['raw'], names)
    err, where, k = rows[0]
    hist.append(err)
    print(f"trial {trial:3d} split {int(S.WINO_SPLIT_BF16)}: worst {err:.2e} @ {where}; next {rows[1][0]:.2e} @ {rows[1][1]}   "
          f"[{time.time() - t_start:.0f} s]", flush=True)
    if err > flag:
        r32 = R.RefCRNN(64, T, 1, 9).to(dev).load_from(model)
        q32 = R.reference_step(r32, feats, y, decisions=d)
        stock = rel_rows(q32['raw'], q['raw'], names)
        stock_here = [s for s in stock if s[2] == k][0][0]
        dev_ = (g[k].double() - q['raw'][k]).abs()
        peak = float(q['raw'][k].abs().max())
        flat = dev_.flatten()
        top = torch.topk(flat, min(3, flat.numel()))
        line = (f"   FLAGGED {where} shape {tuple(g[k].shape)}: product {err:.2e}, STOCK fp32 (same decisions) on this tensor {stock_here:.2e}, "
                f"stock's own worst {stock[0][0]:.2e} @ {stock[0][1]}; gradient peak {peak:.3e}, rms {float(q['raw'][k].pow(2).mean().sqrt()):.3e}")
        if g[k].dim() == 4:
            per_out = dev_.pow(2).sum(dim=(1, 2, 3))
            line += (f"; deviation energy in the top output channel {float(per_out.max() / per_out.sum()):.2f} (1 / {g[k].shape[0]} = {1 / g[k].shape[0]:.2f} if diffuse)"
                     f"; median / max |dev| {float(flat.median()):.2e} / {float(flat.max()):.2e}")
        line += "; top deviations at " + ", ".join(str(tuple(int(v) for v in torch.unravel_index(i, g[k].shape))) + f" {float(v) / peak:.1e}" for v, i in zip(top.values, top.indices))
        print(line, flush=True)
        del r32, q32
    del r64, q, d, m, g
h = torch.tensor(hist)
print(f"{trials} trials: median {float(h.median()):.2e}, max {float(h.max()):.2e}, above 2.5e-5: {int((h > 2.5e-5).sum())}, above 5e-5: {int((h > 5e-5).sum())}")
