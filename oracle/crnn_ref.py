"""Reference for the CRNN half of the path -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement, from the reference's layer definitions, of

    define_keras_model (v = 9 and the plain variants)   sj_train.py:214-255  (ConvMPBlock :191-201, FullyConnectedLayer :204-211)
    BinaryCrossentropy                                  sj_train.py:447-448
    adaptive_clip_grad + unitwise_norm                  sj_train.py:145-155, utils.py:350-366
    Adam(clipvalue)                                     sj_train.py:434-435

with STOCK torch layers only (nn.Conv2d / BatchNorm / ReLU / MaxPool2d / Linear / LSTM): none of this repository's modules,
autograd functions or HIP kernels is imported here, so a comparison against it is a comparison of the product's step with
"the same network on plain PyTorch".  It runs in fp32 (what the reference trains in) and, after `.double()`, in fp64 - the
yardstick both the HIP passes and the stock fp32 ops are measured against at the full c3 / c4 size, where an fp32-vs-fp32
comparison cannot separate a wrong kernel from a ReLU / max-pool decision that two correct fp32 evaluations make differently.

Only `tests/`, `__graft_entry__.smoke()` and bench.py's checker legs may import this module; the product never does.

Pinning: the layer semantics are Keras' (Conv2D 'same', BatchNormalization momentum 0.99 / epsilon 1e-3, MaxPooling2D 'same',
TimeDistributed(Dense), Bidirectional(LSTM)); the reference has no numeric test of its model, so this is
pinned by the layer walk of `tests/test_host.py` (a NumPy forward written from the Keras layer definitions) and by torch's own
operators, not by reference outputs: TensorFlow cannot be imported here.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class RefCRNN(nn.Module):
    """Input x [B, n_mels, n_frame, n_chan] (the reference's channels-last Keras input), output [B, n_frame / 32, 3].

    Parameters and buffers are registered in the layer order of define_keras_model, which is also the order of the
    product's `CustomModel.parameters()` / `.buffers()`: `load_from` copies by position and checks every shape."""

    def __init__(self, n_mels: int, n_frame: int, n_chan: int, v: int = 9, model_type: str = 'vad'):
        super().__init__()
        if model_type == 'vad' and v in (6, 7):
            raise NotImplementedError("RefCRNN: the smoothing-pool (v6) / bottleneck (v7) variants are not restated")
        fsize = 48 if (model_type == 'vad' and v == 8) else 32                       # :215-217
        self.blocks = nn.ModuleList()
        cin = n_chan
        for i in range(5):                                                           # :222-242
            cout, layers = fsize * 2 ** i, nn.ModuleList()
            for _ in range(2 if i == 0 else 3):
                layers.append(nn.ModuleList([nn.Conv2d(cin, cout, 3, padding=1),     # Conv2D(fsize, 3, padding='same')
                                             nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)]))   # Keras: momentum 0.99, epsilon 1e-3
                cin = cout
            self.blocks.append(layers)
        m_out = n_mels
        for _ in range(5):
            m_out = -(-m_out // 2)                                                   # MaxPooling2D 'same': ceil
        v9 = model_type == 'vad' and v == 9
        self.td = nn.Linear(m_out * cin, 1024)                                       # :245
        dims = ([512] if v9 else []) + [256, 128]                                    # :246-249
        self.fc_pre = nn.ModuleList()
        d = 1024
        for n in dims:
            self.fc_pre.append(nn.ModuleList([nn.Linear(d, n), nn.BatchNorm1d(n, eps=1e-3, momentum=0.01)]))
            d = n
        self.lstm = nn.LSTM(128, 128, batch_first=True, bidirectional=True) if v9 else None   # :250-251
        self.fc_post = nn.ModuleList([nn.Linear(256 if v9 else 128, 64), nn.BatchNorm1d(64, eps=1e-3, momentum=0.01)])
        self.head = nn.Linear(64, 3)                                                 # :253, sigmoid for 'vad'
        self.sigmoid_head = model_type == 'vad'
        self.pre_activation = None   # the head's Dense output of the last forward (before the sigmoid)

    @staticmethod
    def _dense_bn_relu(x, fc, bn, mask=None):
        z = fc(x)                                           # [B, T, C]; BatchNormalization over the last axis
        z = bn(z.transpose(1, 2)).transpose(1, 2)
        return torch.relu(z) if mask is None else z * mask

    def forward(self, x, decisions: "Decisions" = None):
        """`decisions`: take every ReLU / max-pool decision from there (what another evaluation of the same network decided)
        instead of from this evaluation's own values - see `Decisions`."""
        d = decisions
        x = x.permute(0, 3, 1, 2)                           # [B, C, M, T]
        k = 0
        for b, layers in enumerate(self.blocks):
            for conv, bn in layers:
                x = bn(conv(x))
                x = torch.relu(x) if d is None else x * d.conv_masks[k]
                k += 1
            if d is None:
                x = nn.functional.max_pool2d(x, 2, 2, ceil_mode=True)    # 'same' pooling pads at the far edge only
            else:
                x = _windows(x).gather(-1, d.pool_slots[b].unsqueeze(-1)).squeeze(-1)
        x = x.permute(0, 3, 2, 1).flatten(2)                # Permute((2, 1, 3)) + Reshape: [B, T', M' * C], m' major  (:243-244)
        x = self.td(x)
        x = torch.relu(x) if d is None else x * d.td_mask
        fcs = list(self.fc_pre)
        for j, (fc, bn) in enumerate(fcs):
            x = self._dense_bn_relu(x, fc, bn, None if d is None else d.fc_masks[j])
        if self.lstm is not None:
            x, _ = self.lstm(x)
        x = self._dense_bn_relu(x, *self.fc_post, None if d is None else d.fc_masks[len(fcs)])
        z = self.head(x)
        self.pre_activation = z
        return torch.sigmoid(z) if self.sigmoid_head else torch.relu(z)

    @torch.no_grad()
    def load_from(self, model: nn.Module) -> "RefCRNN":
        """Copy every parameter and buffer of a product `CustomModel` (or of another RefCRNN) by position."""
        mine, theirs = list(self.parameters()), list(model.parameters())
        if len(mine) != len(theirs):
            raise ValueError(f"RefCRNN.load_from: {len(theirs)} parameters given, {len(mine)} expected")
        for k, (a, b) in enumerate(zip(mine, theirs)):
            if tuple(a.shape) != tuple(b.shape):
                raise ValueError(f"RefCRNN.load_from: parameter {k} is {tuple(b.shape)}, expected {tuple(a.shape)}")
            a.copy_(b.detach().to(a.dtype))
        mine, theirs = list(self.buffers()), list(model.buffers())
        if len(mine) != len(theirs):
            raise ValueError(f"RefCRNN.load_from: {len(theirs)} buffers given, {len(mine)} expected")
        for k, (a, b) in enumerate(zip(mine, theirs)):
            if tuple(a.shape) != tuple(b.shape):
                raise ValueError(f"RefCRNN.load_from: buffer {k} is {tuple(b.shape)}, expected {tuple(a.shape)}")
            a.copy_(b.detach().to(a.dtype))
        return self


def _windows(x):
    """[B, C, H, W] -> [B, C, ceil(H/2), ceil(W/2), 4]: the 2x2 / stride 2 / 'same' pooling windows in (h, w) scan order; the
    elements a window at the far edge lacks are filled with -1 (below every ReLU output)."""
    b, c, h, w = x.shape
    if h % 2 or w % 2:
        x = nn.functional.pad(x, (0, w % 2, 0, h % 2), value=-1.0)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    return x.reshape(b, c, ho, 2, wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, ho, wo, 4)


class Decisions:
    """The discrete half of one training-mode forward of the PRODUCT's model: which units every ReLU passed and which element
    every max-pool window took.

    Why: the network is piecewise smooth.  Two correct fp32 evaluations (this repository's HIP passes, the stock torch / MIOpen
    ops) differ by ~1e-6 in their activations; at batch 64 x 512 frames ~1e8 units sit behind a ReLU or in a pooling window,
    and a few dozen of them lie within that 1e-6 of their decision boundary.  Each such unit taken the other way moves a
    weight gradient (a sum with heavy cancellation under BatchNorm) by ~1e-2 of its peak: profiles/r6/fullsize_parity.log
    shows the stock fp32 ops 2.7e-2 from the fp64 reference, exactly as far as the HIP step.  A gradient comparison at that
    size therefore says nothing unless both sides take the SAME decisions.  With them fixed, the step is a smooth function and
    its gradients can be held to 2e-5 of their peak against fp64 (tests/test_fullsize_gpu.py).

    Built from `hip_autograd.record_activations()` (one entry per fused BatchNorm pass: z, y, statistics) plus the TimeDistributed
    Dense's pre-activation.  A plain layer's mask is y > 0.  A block's last layer writes only the POOLED y; its full-resolution
    activation is re-derived from z with the kernel's own arithmetic, y = max(fma(z, gamma rstd, beta - mean gamma rstd), 0) in
    fp32 (k_elementwise.h `k_bn_relu_pool_apply`), evaluated here through fp64 (the product z sc is exact in fp64; rounding the sum
    once more to fp32 reproduces the fma up to a 2^-29 double-rounding event) - and CHECKED: max-pooling the re-derived
    activation must give the kernel's pooled output bit for bit, otherwise this raises."""

    def __init__(self, tap, td_pre):
        self.conv_masks, self.pool_slots, self.fc_masks = [], [], []
        self.td_mask = td_pre.detach() > 0
        self.rederived = 0
        for e in tap:
            y = e['y']
            if len(self.pool_slots) == 5:   # behind the five blocks: the Dense + BatchNorm + ReLU layers
                self.fc_masks.append((y > 0).squeeze(-1).permute(0, 2, 1))          # Dense + BatchNorm + ReLU on [B, C, T, 1]
                continue
            if not e['pool']:
                self.conv_masks.append(y > 0)
                continue
            full = self._rederive(e)
            self.conv_masks.append(full > 0)
            self.pool_slots.append(_windows(full).argmax(-1))                       # first maximum in scan order, as the kernel
            self.rederived += 1

    @staticmethod
    def _rederive(e):
        z, y = e['z'], e['y']
        c = z.shape[1]
        sc = (e['gamma'] * e['rstd']).float()
        shifts = (e['beta'] - e['mean'] * sc,                                                        # separately rounded
                  (e['beta'].double() - e['mean'].double() * sc.double()).float())                   # contracted to one fma
        for sh in shifts:
            full = (z.double() * sc.double().view(1, c, 1, 1) + sh.double().view(1, c, 1, 1)).float().clamp_min_(0.0)
            if torch.equal(nn.functional.max_pool2d(full, 2, 2, ceil_mode=True), y):
                return full
        raise AssertionError("Decisions: the pooled layer's activation re-derived from z does not reproduce the kernel's pooled "
                             "output bit for bit - the kernel's arithmetic has changed; update Decisions._rederive with it")


def binary_crossentropy(y_true, y_pred):
    """tf.keras.losses.BinaryCrossentropy() (sj_train.py:447-448): probabilities clipped to [1e-7, 1 - 1e-7], mean over everything."""
    p = y_pred.clamp(1e-7, 1 - 1e-7)
    return -(y_true * p.log() + (1 - y_true) * (1 - p).log()).mean()


def unitwise_norm(t):
    """utils.py:350-366 in torch's layouts: vectors -> one norm; Dense [out, in] / LSTM [4u, in] -> per row (Keras [in, out]
    axis 0); conv [out, in, kh, kw] -> per output channel (Keras HWIO axes 0, 1, 2)."""
    if t.dim() <= 1:
        return (t ** 2).sum() ** 0.5
    return (t ** 2).sum(dim=tuple(range(1, t.dim())), keepdim=True) ** 0.5


def adaptive_clip_grad(parameters, gradients, clip_factor=0.01, eps=1e-3):
    """sj_train.py:145-155."""
    out = []
    for p, g in zip(parameters, gradients):
        max_norm = unitwise_norm(p).clamp(min=eps) * clip_factor
        g_norm = unitwise_norm(g)
        out.append(torch.where(g_norm < max_norm, g, g * (max_norm / g_norm.clamp(min=1e-6))))
    return out


def reference_step(ref: RefCRNN, x, y, clipvalue=0.01, use_agc=True, decisions: Decisions = None):
    """One training-mode forward / backward of `ref` on (x, y) as CustomModel.train_step does it (sj_train.py:162-182) up to
    the optimiser: returns loss, outputs, the raw gradients, and the gradients after AGC + Adam's element-wise clipvalue.
    BatchNorm running statistics of `ref` are updated (training mode), parameters are not."""
    ref.train()
    for p in ref.parameters():
        p.grad = None
    x = x.to(next(ref.parameters()).dtype)
    y = y.to(x.dtype)
    out = ref(x, decisions)
    loss = binary_crossentropy(y, out)
    loss.backward()
    params = list(ref.parameters())
    raw = [p.grad.detach().clone() for p in params]
    clipped = adaptive_clip_grad([p.detach() for p in params], raw) if use_agc else [g.clone() for g in raw]
    if clipvalue:
        clipped = [g.clamp(-clipvalue, clipvalue) for g in clipped]
    return {"loss": loss.detach(), "out": out.detach(), "pre_activation": ref.pre_activation.detach(), "raw": raw, "clipped": clipped}
