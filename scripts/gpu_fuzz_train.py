"""Randomised sweep of the CRNN-side HIP passes against the stock torch / MIOpen ops: ConvMPBlock (BatchNorm + ReLU + MaxPool
passes, the first-layer form with 1-2 input channels), FullyConnectedLayer, the bidirectional LSTM - outputs and every gradient.
usage: gpu_fuzz_train.py [n_cases] [seed] [--split]      (--split: the HIP side runs its Winograd passes on the BF16 matrix cores)"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_cases = int(_args[0]) if len(_args) > 0 else 40
rng = np.random.default_rng(int(_args[1]) if len(_args) > 1 else 0)
SPLIT = "--split" in sys.argv
FLAGS = ("FUSED_BN_RELU", "FUSED_BN_POOL", "FUSED_BN_STATS", "FUSED_CONV0", "FUSED_LSTM", "FUSED_FC_BN", "WINO_TRAIN", "WINO_TRAIN_WRW",
         "C32_TRAIN", "ZERO_POOL")


def set_flags(v):
    for f in FLAGS:
        setattr(S, f, v)
    S.WINO_SPLIT_BF16 = bool(v) and SPLIT


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-6)


bad = flips = 0
for case in range(n_cases):
    kind = rng.choice(["block", "block", "fc", "lstm"])
    torch.manual_seed(int(rng.integers(1 << 30)))
    if kind == "block":
        cin = int(rng.choice([1, 2, 8, 32, 64, 64, 128, 256]))
        cout = int(rng.choice([8, 16, 32, 64, 64, 128, 256]))
        # (B >= 2: the STOCK side of the comparison - torch's BatchNorm2d on MIOpen, channels-last, training mode - dumps core on
        # batch-1 inputs in this image)
        nconv, b, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 6)), int(rng.integers(2, 20)), int(rng.integers(2, 70))
        mod = S.ConvMPBlock(cin, num_convs=nconv, fsize=cout, BN=True, MP=bool(rng.random() < 0.8)).to(dev).to(memory_format=torch.channels_last).train()
        x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        needs_grad = bool(rng.random() < 0.5)     # False: the first layer takes the recomputing form when cin <= 2
        desc = f"block cin {cin} cout {cout} convs {nconv} B {b} H {h} W {w} pool {isinstance(mod.pool, torch.nn.MaxPool2d)} xgrad {needs_grad}"
    elif kind == "fc":
        cin, cout = int(rng.choice([16, 64, 256, 1024])), int(rng.choice([8, 64, 128, 512]))
        b, t = int(rng.integers(1, 9)), int(rng.integers(2, 40))
        mod = S.FullyConnectedLayer(cin, cout, BN=True).to(dev).train()
        x = torch.randn(b, t, cin, device=dev)
        needs_grad = True
        desc = f"fc {cin} -> {cout} B {b} T {t}"
    else:
        b, t = int(rng.integers(1, 70)), int(rng.integers(1, 50))
        lstm = torch.nn.LSTM(128, 128, batch_first=True, bidirectional=True).to(dev)

        class Wrap(torch.nn.Module):
            def __init__(self, l):
                super().__init__()
                self.l = l

            def forward(self, v):
                return S.bilstm128(self.l, v) if S.FUSED_LSTM else self.l(v)[0]
        mod = Wrap(lstm)
        x = torch.randn(b, t, 128, device=dev) * float(rng.choice([0.3, 1.0, 3.0]))
        needs_grad = True
        desc = f"lstm B {b} T {t}"
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.weight.uniform_(-1.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(mod)
    xa, xb = x.clone().requires_grad_(needs_grad), x.clone().requires_grad_(needs_grad)
    set_flags(True)
    ya = mod(xa)
    g = torch.randn_like(ya)
    ya.backward(g)
    set_flags(False)
    yb = ref(xb)
    yb.backward(g)
    errs = {"out": rel(ya, yb)}
    if needs_grad:
        errs["dx"] = rel(xa.grad, xb.grad)
    for (n, pa), (_, pb) in zip(mod.named_parameters(), ref.named_parameters()):
        if pb.grad is None or (kind != "lstm" and (n.endswith("0.bias") or n == "fc.bias")):  # bias in front of a BatchNorm: zero by construction
            continue
        errs[n] = rel(pa.grad, pb.grad)
    for (n, ba), (_, bb) in zip(mod.named_buffers(), ref.named_buffers()):
        if ba.dtype.is_floating_point:
            errs["buf " + n] = rel(ba, bb)
    worst = max(errs, key=errs.get)
    # A ReLU mask / pooling winner decided within rounding flips one gradient contribution: an outlier of 1e-3..1e-2 on a
    # small layer.  The stock ops do the same against themselves (x vs x (1 + 1e-7): 5e-7 typically, 8e-3 once in 24 runs of
    # one shape), so such a case counts as a flip when the forward outputs still agree, not as a failure.
    ok = errs[worst] <= 2e-4
    flip = (not ok) and errs["out"] <= 1e-5 and errs[worst] <= 2e-2
    bad += not (ok or flip)
    flips += flip
    print(("ok  " if ok else "flip" if flip else "FAIL"), desc, f"worst {worst} {errs[worst]:.1e}")
    if not (ok or flip):  # diagnostics: is it a mask decided within rounding?  the stock ops against themselves on x (1 + 1e-7)
        ref2 = copy.deepcopy(ref)
        for q in ref2.parameters():
            q.grad = None
        xc = (x * (1 + 1e-7)).clone().requires_grad_(needs_grad)
        set_flags(False)
        yc = ref2(xc)
        yc.backward(g)
        self_err = max(rel(pc.grad, pb.grad) for (n, pc), (_, pb) in zip(ref2.named_parameters(), ref.named_parameters())
                       if pb.grad is not None and not (n.endswith("0.bias") or n == "fc.bias"))
        # ... or directly: do the two sides take a different ReLU / pooling decision anywhere?  (the HIP side's activations from
        # hip_autograd.record_activations, a pooled layer's full-resolution activation re-derived and verified as in
        # oracle.crnn_ref.Decisions; the stock side's from hooks on its BatchNorm layers)
        differing = None
        if kind == "block":
            from challenge_amd.hip_autograd import record_activations
            from oracle import crnn_ref as RR
            set_flags(True)
            with torch.no_grad(), record_activations() as acts:
                copy.deepcopy(ref).train()(x)
            stock_full = []
            probe = copy.deepcopy(ref).train()
            hooks = [m.register_forward_hook(lambda mod, i, o: stock_full.append(torch.relu(o.detach())))
                     for m in probe.modules() if isinstance(m, torch.nn.BatchNorm2d)]
            set_flags(False)
            with torch.no_grad():
                probe(x)
            for h in hooks:
                h.remove()
            differing = 0
            for e, full_b in zip(acts, stock_full):
                full_a = RR.Decisions._rederive(e) if e['pool'] else e['y']
                differing += int(((full_a > 0) != (full_b > 0)).sum())
                if e['pool']:
                    differing += int((RR._windows(full_a).argmax(-1) != RR._windows(full_b).argmax(-1)).sum())
            if errs["out"] <= 1e-5 and differing > 0:
                bad -= 1
                flips += 1
                print(f"     (reclassified as a flip: {differing} ReLU / pooling decisions differ between the two sides while the outputs agree)")
        if errs["out"] <= 1e-5 and self_err >= 0.5 * errs[worst] and not differing:
            # the stock ops, perturbed in the last bit of their input, move their own gradients as far: a decision made within
            # rounding with a large gradient behind it - a flip after all, just above the 2e-2 bound
            bad -= 1
            flips += 1
            print("     (reclassified as a flip: the stock ops against themselves deviate as much)")
        print(f"     out {errs['out']:.1e}; outputs with a different sign of activity: {int(((ya > 0) != (yb > 0)).sum())} of {ya.numel()}; "
              f"stock ops vs themselves on x (1 + 1e-7): worst gradient {self_err:.1e}; all: " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items() if v > 2e-4))
print("failures:", bad, " flips:", flips)
sys.exit(1 if bad else 0)
