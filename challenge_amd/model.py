"""The CRNN of the reference's define_keras_model (sj_train.py:191-255 upstream) as a torch nn.Module with the Keras-style
training surface the reference uses (compile / train_step :158-188 / test_step / predict), its inference-only execution
(`InferenceEngine`: BatchNorm folded, every convolution a HIP kernel, frontend + forward as one hipGraph) and the import of a
reference checkpoint (`load_keras_weights`).  Which HIP pass a layer takes is decided here per call from switches.py; the
passes themselves live in hip_autograd.py."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import os
import torch
import torch.nn as nn

from . import frontend as _fe
from . import switches as SW
from .hip_autograd import (FusedAGC, _FusedBiasBNReLU, _FusedConv0BNReLU, _IN_STEP, _PACKS, _WinoConv3x3, _ZERO_POOL, _is_first_layer_conv,
                           _is_pool_2x2_same, _lstm_is_bilstm128, _wino_train_conv, adaptive_clip_grad, bilstm128)


# BatchNorm's num_batches_tracked counters of the layers whose fused passes ran, bumped by ONE _foreach_add_ at the end of
# CustomModel.forward instead of one launch per layer (None outside that forward: the layers then bump their own)
_NBT_PENDING = None


def _count_batch(bn) -> None:
    if _NBT_PENDING is not None:
        _NBT_PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


class _ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout, k=3, bn=True):
        layers = [nn.Conv2d(cin, cout, k, padding=k // 2)]
        if bn:
            layers.append(nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01))  # Keras BN defaults
        layers.append(nn.ReLU(inplace=True))
        super().__init__(*layers)

    def forward(self, x, pool=None):
        """`pool`: the MaxPool2d(2, 2, ceil_mode=True) that follows this layer in its ConvMPBlock (applied here, inside the
        fused passes when they run, as the module otherwise)."""
        if (SW.FUSED_BN_RELU and self.training and x.is_cuda and len(self) == 3 and isinstance(self[1], nn.BatchNorm2d)
                and not torch.is_autocast_enabled()):
            conv, bn = self[0], self[1]
            if x.dtype == torch.float32 and conv.out_channels % 4 == 0 and bn.track_running_stats and bn.momentum is not None:
                if SW.FUSED_CONV0 and pool is None and _is_first_layer_conv(conv, x):
                    _count_batch(bn)  # the model's first layer: convolution recomputed inside the passes
                    return _FusedConv0BNReLU.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                                   bn.running_var, bn.eps, bn.momentum)
                wino = _wino_train_conv(conv, x)
                sums0 = None
                if wino is not None and wino[0] and SW.FUSED_BN_STATS:
                    # a HIP kernel runs the forward: the BatchNorm's statistics come out of its epilogue (no pass over z for them)
                    z, sums0 = _WinoConv3x3.apply(x, conv.weight, wino[0], wino[1], wino[2], True)
                elif wino is not None:
                    z = _WinoConv3x3.apply(x, conv.weight, wino[0], wino[1], wino[2])
                else:
                    z = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
                if z.is_contiguous(memory_format=torch.channels_last):
                    _count_batch(bn)
                    fold = SW.FUSED_BN_POOL and _is_pool_2x2_same(pool)
                    y = _FusedBiasBNReLU.apply(z, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                               bn.eps, bn.momentum, fold, sums0)
                    return y if (fold or pool is None) else pool(y)
                y = self[2](bn(z + conv.bias.view(1, -1, 1, 1) if conv.bias is not None else z))
                return y if pool is None else pool(y)
        y = super().forward(x)
        return y if pool is None else pool(y)


class ConvMPBlock(nn.Module):
    """num_convs x [Conv3x3 'same' (+BN) + ReLU] + MaxPool 2x2 'same' (sj_train.py:191-201)."""

    def __init__(self, cin, num_convs=2, fsize=32, kernel_size=3, BN=False, MP=True):
        super().__init__()
        self.convs = nn.Sequential(*[_ConvBNReLU(cin if i == 0 else fsize, fsize, kernel_size, BN)
                                     for i in range(num_convs)])
        self.pool = nn.MaxPool2d(2, 2, ceil_mode=True) if MP else nn.Identity()

    def forward(self, x):
        layers = list(self.convs)
        for layer in layers[:-1]:
            x = layer(x)
        if isinstance(layers[-1], _ConvBNReLU) and isinstance(self.pool, nn.MaxPool2d):
            return layers[-1](x, pool=self.pool)  # the pooling goes into the last layer's fused passes when those run
        return self.pool(layers[-1](x))


class FullyConnectedLayer(nn.Module):
    """Dense (+BN over the feature axis) + activation on [B, T, units] (sj_train.py:204-211)."""

    def __init__(self, cin, nodes=512, act='relu', BN=False):
        super().__init__()
        self.fc = nn.Linear(cin, nodes)
        self.bn = nn.BatchNorm1d(nodes, eps=1e-3, momentum=0.01) if BN else None
        self.act = {'relu': nn.ReLU(inplace=True), 'sigmoid': nn.Sigmoid()}[act]

    def forward(self, x):
        bn = self.bn
        if (SW.FUSED_BN_RELU and SW.FUSED_FC_BN and self.training and bn is not None and isinstance(self.act, nn.ReLU) and x.is_cuda
                and x.dim() == 3 and x.dtype == torch.float32 and self.fc.out_features % 4 == 0 and bn.track_running_stats
                and bn.momentum is not None and not torch.is_autocast_enabled()):
            # Dense + BatchNorm over the feature axis + ReLU = the convolution layers' passes on a [rows = B T, C] activation:
            # the GEMM runs without bias (BatchNorm removes it), no transposes, no separate normalise / ReLU kernels
            z = torch.nn.functional.linear(x, self.fc.weight)                # [B, T, C]
            z4 = z.permute(0, 2, 1).unsqueeze(-1)                             # [B, C, T, 1]: a channels_last view of the same memory
            if z4.is_contiguous(memory_format=torch.channels_last):
                _count_batch(bn)
                y4 = _FusedBiasBNReLU.apply(z4, self.fc.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                            bn.momentum, False)
                return y4.squeeze(-1).permute(0, 2, 1)
            x = z + self.fc.bias if self.fc.bias is not None else z
        elif (bn is None and not self.training and isinstance(self.act, nn.ReLU) and x.is_cuda and x.dim() == 3
              and not torch.is_grad_enabled() and self.fc.bias is not None and not torch.is_autocast_enabled()):
            # inference with the BatchNorm folded away: Dense + bias + ReLU as ONE GEMM with a fused epilogue
            b, t, c = x.shape
            return torch._addmm_activation(self.fc.bias, x.reshape(b * t, c), self.fc.weight.t()).view(b, t, -1)
        else:
            x = self.fc(x)
        if bn is not None:
            x = bn(x.transpose(1, 2)).transpose(1, 2)
        return self.act(x)


class _Bottleneck(nn.Module):  # v == 7 residual block (sj_train.py:230-241)
    def __init__(self, c):
        super().__init__()
        self.body = nn.Sequential(_ConvBNReLU(c, c // 4, 1), _ConvBNReLU(c // 4, c // 4, 3), _ConvBNReLU(c // 4, c, 1))

    def forward(self, x):
        return self.body(x) + x


class _SmoothPool(nn.Module):  # v == 6 (sj_train.py:225-229): avg (1,k) then max (1,2k), stride 1, 'same'
    def __init__(self, k):
        super().__init__()
        self.k = max(int(k), 1)

    @staticmethod
    def _same(x, k, mode):
        if k <= 1:
            return x
        left = (k - 1) // 2
        right = k - 1 - left
        F = torch.nn.functional
        if mode == 'max':
            return F.max_pool2d(F.pad(x, (left, right), value=float('-inf')), (1, k), 1)
        ones = torch.ones((1, 1, 1, x.shape[-1]), dtype=x.dtype, device=x.device)
        cnt = F.avg_pool2d(F.pad(ones, (left, right)), (1, k), 1)  # valid fraction, as TF 'SAME' averages
        return F.avg_pool2d(F.pad(x, (left, right)), (1, k), 1) / cnt

    def forward(self, x):
        return self._same(self._same(x, self.k, 'avg'), 2 * self.k, 'max')


class CustomModel(nn.Module):
    """The CRNN of define_keras_model plus the Keras-style training surface the reference
    uses: compile(), train_step(data) (sj_train.py:158-188), test_step, fit."""

    def __init__(self, config):
        super().__init__()
        fsize = 48 if (config.model_type == 'vad' and config.v == 8) else 32
        self.config_v, self.model_type = config.v, config.model_type
        blocks = [ConvMPBlock(config.n_chan, 2, fsize, BN=True)]
        cin, width = fsize, config.n_frame // 2
        for i in range(1, 5):
            if config.model_type == 'vad' and config.v == 6:
                k = int(round(0.5 / (256 * config.n_frame / 16000 / width)))
                blocks.append(_SmoothPool(k))
            if config.model_type == 'vad' and config.v == 7:
                blocks.append(_Bottleneck(cin))
            blocks.append(ConvMPBlock(cin, 3, fsize * 2 ** i, BN=True))
            cin, width = fsize * 2 ** i, -(-width // 2)
        self.features = nn.Sequential(*blocks)
        m_out = config.n_mels
        for _ in range(5):
            m_out = -(-m_out // 2)
        v9 = config.model_type == 'vad' and config.v == 9
        self.td = nn.Linear(m_out * cin, 1024)
        fcs, d = [], 1024
        if v9:
            fcs.append(FullyConnectedLayer(d, 512, BN=True)); d = 512
        fcs.append(FullyConnectedLayer(d, 256, BN=True))
        fcs.append(FullyConnectedLayer(256, 128, BN=True))
        self.fc_pre = nn.Sequential(*fcs)
        self.lstm = nn.LSTM(128, 128, batch_first=True, bidirectional=True) if v9 else None
        self.fc_post = FullyConnectedLayer(256 if v9 else 128, 64, BN=True)
        self.head = FullyConnectedLayer(64, 3, act='sigmoid' if config.model_type == 'vad' else 'relu')
        self.optimizer = None
        self.loss_fn: Optional[Callable] = None
        self.clipvalue: Optional[float] = None
        self.use_agc = True
        object.__setattr__(self, '_ddp', None)  # not a submodule: DDP wraps this very module
        object.__setattr__(self, '_fused_agc', None)
        # bumped by everything that changes parameters or buffers WITHOUT going through ATen's version counters: the raw-
        # pointer BatchNorm / AGC kernels, hipGraph replays (GraphedTrainStep), load_state_dict; `predict` keys its cached
        # InferenceEngine on it
        object.__setattr__(self, '_generation', 0)

    def forward(self, x):
        """x: [B, n_mels, n_frame, n_chan] (the reference's channels-last input)."""
        global _NBT_PENDING
        if SW.ZERO_POOL and not _IN_STEP[0] and self.training and x.is_cuda and torch.is_grad_enabled():
            # a training-mode pass outside train_step (a custom loop, a test's grads()): it is its own "step" for the zero pool,
            # whose demand would otherwise add up over such passes until the next train_step allocated twice their SUM
            _ZERO_POOL.begin_step(x.device)
        if SW.FUSED_PACK and SW.WINO_TRAIN and self.training and x.is_cuda and torch.is_grad_enabled():
            _PACKS.prepack(x.device)   # every Winograd weight packing this step will ask for, in one launch (hip_autograd._PackBook)
        outer, _NBT_PENDING = _NBT_PENDING, []
        try:
            return self._forward(x)
        finally:
            pending, _NBT_PENDING = _NBT_PENDING, outer
            if pending:
                torch._foreach_add_(pending, 1)

    def _forward(self, x):
        x = x.permute(0, 3, 1, 2)  # NCHW view of the NHWC tensor (channels_last strides)
        x = self.features(x)       # [B, C, M', T']
        x = x.permute(0, 3, 2, 1).flatten(2)  # [B, T', M' * C], m' major as Keras Permute+Reshape
        x = torch.relu(self.td(x))
        x = self.fc_pre(x)
        if self.lstm is not None:
            if (SW.FUSED_LSTM and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
                    and _lstm_is_bilstm128(self.lstm)):
                x = bilstm128(self.lstm, x)  # the recurrence (and its backward through time) in one HIP launch each
            else:
                x, _ = self.lstm(x)
        return self.head(self.fc_post(x))

    # ---- Keras-like training surface ------------------------------------
    def compile(self, optimizer, loss, clipvalue: Optional[float] = None, use_agc: bool = True, ddp=None):
        self.optimizer, self.loss_fn, self.clipvalue, self.use_agc = optimizer, loss, clipvalue, use_agc
        object.__setattr__(self, '_ddp', ddp)

    def _call(self, x):
        return self._ddp(x) if self._ddp is not None else self(x)

    def train_step(self, data, _mark=None):
        """Forward, loss, backward, AGC on the (all-reduced) gradients, element-wise
        clipvalue, optimiser step (sj_train.py:162-188).  Returns {'loss': tensor}.
        `_mark(name)` (bench hook) is called after each phase: 'forward', 'backward', 'agc_clip', 'optimizer'."""
        mark = _mark or (lambda name: None)
        x, y = data
        if not self.training:   # (Module.train() walks every submodule: 0.4 ms of host time per step when nothing changes)
            self.train()
        self.bump_generation()
        # the gradients are dropped, not zeroed: AccumulateGrad then takes each incoming gradient instead of adding it to a
        # zeroed buffer (84 elementwise launches and the zero fills fewer per step); FusedAGC keeps one table per recurring
        # address set of the gradient buffers, so nothing is re-uploaded in steady state
        self.optimizer.zero_grad(set_to_none=True)
        if x.is_cuda and SW.ZERO_POOL:
            _ZERO_POOL.begin_step(x.device)
        was_in_step, _IN_STEP[0] = _IN_STEP[0], True
        try:
            y_pred = self._call(x)
        finally:
            _IN_STEP[0] = was_in_step
        loss = self.loss_fn(y, y_pred)
        mark('forward')
        loss.backward()  # under DDP the bucketed RCCL all-reduce overlaps with this
        mark('backward')
        fused = self.use_agc and x.is_cuda  # one HIP launch for AGC + clipvalue over the whole model
        stepped = False
        if fused:
            if self._fused_agc is None:
                object.__setattr__(self, '_fused_agc', FusedAGC(list(self.parameters())))
            # AGC + clipvalue, and - for the plain Adam make_optimizer builds - the optimiser's update in the same launch
            if self._fused_agc.attach_adam(self.optimizer):
                stepped = self._fused_agc.adam_step(0.01, 1e-3, self.clipvalue)
            if not stepped:
                self._fused_agc(0.01, 1e-3, self.clipvalue)
        else:
            params = [p for p in self.parameters() if p.grad is not None]
            if self.use_agc:
                new = adaptive_clip_grad(params, [p.grad for p in params])
                for p, g in zip(params, new):
                    p.grad = g
            if self.clipvalue:
                torch.nn.utils.clip_grad_value_(params, self.clipvalue)
        mark('agc_clip')
        if not stepped:
            self.optimizer.step()
        mark('optimizer')
        return {'loss': loss.detach()}

    @torch.no_grad()
    def test_step(self, data):
        x, y = data
        self.eval()
        return {'loss': self.loss_fn(y, self(x))}

    def bump_generation(self) -> None:
        """Tell `predict` that parameters / buffers have changed (see `_generation`)."""
        object.__setattr__(self, '_generation', self._generation + 1)
        _PACKS.invalidate()   # (a graph replay updates the weights without touching their autograd version: packed copies are stale)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.bump_generation()
        return out

    def _state_version(self):
        return (self._generation, sum(t._version for t in self.parameters()) + sum(t._version for t in self.buffers()))

    @torch.no_grad()
    def predict(self, x: torch.Tensor, batch_size: int = 32) -> torch.Tensor:
        """Keras `Model.predict` (what metrics.evaluate calls, metrics.py:62): inference in batches of `batch_size`, no
        gradients, training state untouched.  On a GPU it runs through an `InferenceEngine` (BatchNorm folded, HIP epilogues /
        block-1 convolutions / LSTM launch; outputs equal to 1e-4) that is rebuilt whenever a parameter or buffer has changed
        since it was made."""
        if x.is_cuda and x.dtype == torch.float32:
            ver = self._state_version()
            eng = self.__dict__.get('_predict_engine')
            if eng is None or eng[0] != ver:
                was_training = self.training
                eng = (ver, InferenceEngine(self))
                self.train(was_training)
                object.__setattr__(self, '_predict_engine', eng)
            fn = eng[1]
        else:
            was_training = self.training
            self.eval()
            fn = self.__call__
        try:
            outs = [fn(x[i:i + batch_size]) for i in range(0, x.shape[0], batch_size)]
        finally:
            if not (x.is_cuda and x.dtype == torch.float32):
                self.train(was_training)
        return torch.cat(outs) if len(outs) != 1 else outs[0]


@torch.no_grad()
def fold_batchnorm(model: nn.Module) -> nn.Module:
    """Inference-only copy of `model` with every BatchNorm folded into the Conv2d / Linear in front of it
    (eval-mode BN is the affine map y = (x - mean) / sqrt(var + eps) * gamma + beta with fixed statistics:
    W' = W * s, b' = (b - mean) * s + beta, s = gamma / sqrt(var + eps), per output unit).  Same function up to
    fp32 rounding (CPU test: <= 1e-5 on the sigmoid outputs); 18 + 5 normalisation launches fewer per forward of
    the v9 CRNN.  The copy is put in eval mode; training keeps the original (BN needs batch statistics there)."""
    import copy
    keep = {k: model.__dict__.get(k) for k in ('optimizer', '_ddp', '_fused_agc', '_predict_engine')}  # stays with the original
    try:
        for k in keep:
            if k in model.__dict__:
                object.__setattr__(model, k, None)
        m = copy.deepcopy(model).eval()
    finally:
        for k, v in keep.items():
            if k in model.__dict__:
                object.__setattr__(model, k, v)

    def scale_shift(bn):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return s, bn.bias - bn.running_mean * s

    for mod in list(m.modules()):
        if isinstance(mod, _ConvBNReLU) and len(mod) == 3 and isinstance(mod[1], nn.BatchNorm2d):
            conv, bn = mod[0], mod[1]
            s, t = scale_shift(bn)
            conv.weight.mul_(s.view(-1, 1, 1, 1))
            conv.bias.copy_(conv.bias * s + t)
            mod[1] = nn.Identity()
        elif isinstance(mod, FullyConnectedLayer) and mod.bn is not None:
            s, t = scale_shift(mod.bn)
            mod.fc.weight.mul_(s.view(-1, 1))
            mod.fc.bias.copy_(mod.fc.bias * s + t)
            mod.bn = None
    return m


class _ConvBiasReLU(nn.Module):
    """Inference form of a folded _ConvBNReLU: the convolution without bias on MIOpen, then ONE HIP pass for
    bias + ReLU (iris_bias_relu) - or, for the last convolution of a block, bias + ReLU + the block's 2x2 max-pool
    (iris_bias_relu_maxpool) - instead of separate add / clamp / pooling kernels over the activation."""

    def __init__(self, conv: nn.Conv2d, pool: bool, nchw: bool = False, hip: Optional[str] = None):
        """`hip`: None (MIOpen convolution + epilogue pass), 'stencil' (first layer, 1-2 input channels: one-pass HIP stencil,
        channels-last output) or 'mfma32' (32 -> 32 channels: the convolution itself on the fp32 matrix cores with bias, ReLU
        and the block's pooling fused, channels-last in and out)."""
        super().__init__()
        fmt = torch.contiguous_format if (nchw or hip) else torch.channels_last
        self.weight = nn.Parameter(conv.weight.detach().clone(memory_format=fmt), requires_grad=False)
        self.bias = nn.Parameter(conv.bias.detach().clone(), requires_grad=False)
        self.padding, self.pool, self.nchw, self.hip = conv.padding, pool, nchw, hip
        self.out_chunked = False  # 'mfma32' only: hand the Winograd stack behind this layer its chunked activation directly

    @staticmethod
    def hip_form(conv: nn.Conv2d) -> Optional[str]:
        """Which HIP convolution, if any, this (folded) layer's shape has."""
        def pair(v):
            return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        plain = (pair(conv.kernel_size) == (3, 3) and pair(conv.padding) == (1, 1) and pair(conv.stride) == (1, 1)
                 and pair(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None
                 and conv.weight.dtype == torch.float32)
        if plain and conv.in_channels in (1, 2) and conv.out_channels % 4 == 0 and conv.out_channels <= 256 and 1024 % conv.out_channels == 0:
            return 'stencil'
        if plain and conv.in_channels == 32 and conv.out_channels == 32:
            return 'mfma32'
        return None

    def forward(self, x):
        if self.hip == 'stencil' and x.shape[3] <= 2048 and not self.pool:
            b, c, h, w = x.shape  # one channel: channels_last and contiguous coincide in memory
            xc = x.as_strided((b, c, h, w), (h * w, h * w, w, 1)) if (c == 1 and x.stride(3) == 1 and x.stride(2) == w) else x.contiguous()
            return _fe.conv3x3_small_bias_relu(xc, self.weight, self.bias, channels_last=True)
        if self.hip == 'mfma32':
            if not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            return _fe.conv3x3_c32_bias_relu(x, self.weight, self.bias, pool=self.pool, out_chunked=self.out_chunked)
        if self.hip:  # shape outside the HIP kernel's range: MIOpen on the contiguous weight
            y = torch.nn.functional.conv2d(x, self.weight, self.bias, padding=self.padding).relu_()
            return torch.nn.functional.max_pool2d(y, 2, 2, ceil_mode=True) if self.pool else y
        if self.nchw:  # contiguous in, contiguous out - or, with the block's pooling, channels_last out
            if x.shape[1] == 1:  # one channel: NHWC and NCHW coincide in memory; give the view plain NCHW strides, or the
                b, c, h, w = x.shape  # convolution is dispatched as channels_last and its output has to be copied back
                x = x.as_strided((b, c, h, w), (h * w, h * w, w, 1)) if x.stride(3) == 1 and x.stride(2) == w else x.contiguous()
            else:
                x = x.contiguous()
            if (not self.pool and x.shape[1] <= 2 and tuple(self.weight.shape[2:]) == (3, 3) and tuple(self.padding) == (1, 1)
                    and x.shape[3] % 4 == 0 and x.dtype == torch.float32):
                # the model's first layer: a 9 / 18-tap stencil bound by its output stream - convolution, bias, ReLU in one pass
                return _fe.conv3x3_small_bias_relu_nchw(x, self.weight, self.bias)
            y = torch.nn.functional.conv2d(x, self.weight, None, padding=self.padding)
            if not y.is_contiguous():
                y = y.contiguous()
            return _fe.bias_relu_maxpool_nchw(y, self.bias) if self.pool else _fe.bias_relu_nchw_(y, self.bias)
        y = torch.nn.functional.conv2d(x, self.weight, None, padding=self.padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        return _fe.bias_relu_maxpool(y, self.bias) if self.pool else _fe.bias_relu_(y, self.bias)


class _WinoStack(nn.Module):
    """Inference form of a run of ConvMPBlocks with 8 | Cin and 64 | Cout (blocks 2-5 of the CRNN, sj_train.py:222-242): every
    Conv2D 3x3 + folded bias + ReLU (+ the block's MaxPool) as ONE launch of the Winograd F(2x2, 3x3) kernel on the fp32
    matrix cores (iris_conv3x3_wino_bias_relu: 16 instead of 36 multiplies per output tile; MIOpen's implicit GEMMs already
    sit at the fp32 MFMA rate).  The layers hand each other the channel-chunked activation [B, C / 8, H, W, 8]; the first one
    converts from channels_last, the last one writes channels_last again."""

    def __init__(self, blocks):
        super().__init__()
        self.layers = []  # (index, cout, pool, split-bf16 kernel)
        dev = None
        for blk in blocks:
            convs = list(blk.convs)
            has_pool = isinstance(blk.pool, nn.MaxPool2d) or getattr(blk, '_pool_fused', False)
            for i, m in enumerate(convs):
                conv = m[0] if isinstance(m, nn.Sequential) else m
                w, b = conv.weight.detach(), conv.bias.detach()
                dev = w.device
                k = len(self.layers)
                b3 = bool(SW.WINO_SPLIT_BF16) and int(w.shape[1]) % 16 == 0 and w.is_cuda   # BF16 matrix cores, three-term split
                self.register_buffer(f"packed{k}", _fe.wino_pack_weights_device(w.contiguous(), split_bf16=True) if b3
                                     else _fe.wino_pack_weights(w), persistent=False)
                self.register_buffer(f"bias{k}", b.to(torch.float32).contiguous().clone(), persistent=False)
                self.layers.append((k, int(w.shape[0]), has_pool and i == len(convs) - 1, b3))

    @staticmethod
    def eligible(blk) -> bool:
        def pair(v):
            return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        if not isinstance(blk, ConvMPBlock):
            return False
        if not (isinstance(blk.pool, nn.Identity) or _is_pool_2x2_same(blk.pool)):
            return False
        for m in blk.convs:
            conv = m[0] if isinstance(m, nn.Sequential) else m
            if not (isinstance(conv, nn.Conv2d) and (not isinstance(m, nn.Sequential) or (len(m) == 3 and isinstance(m[1], nn.Identity)))):
                return False
            if not (pair(conv.kernel_size) == (3, 3) and pair(conv.padding) == (1, 1) and pair(conv.stride) == (1, 1)
                    and pair(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None
                    and conv.weight.dtype == torch.float32 and conv.in_channels % 8 == 0 and conv.out_channels % 64 == 0):
                return False
        return True

    def forward(self, x):
        if x.dim() != 5:  # (the layer in front may already have written the chunked layout)
            x = _fe.to_chunked(x)
        last = len(self.layers) - 1
        for k, cout, pool, b3 in self.layers:
            x = _fe.conv3x3_wino_bias_relu(x, getattr(self, f"packed{k}"), getattr(self, f"bias{k}"), cout, pool=pool, out_nhwc=(k == last),
                                           split_bf16=b3)
        return x


class _HipBiLSTM(nn.Module):
    """Inference form of the model's nn.LSTM(128, 128, bidirectional, batch_first): ONE GEMM for the input projections of
    all steps and both directions, then the whole recurrence in ONE HIP launch (iris_bilstm128_forward; MIOpen runs a GEMM
    and a pointwise kernel per step and direction).  Returns (output, None) like nn.LSTM."""

    def __init__(self, lstm: nn.LSTM):
        super().__init__()
        if not self.supports(lstm):
            raise ValueError("_HipBiLSTM: a one-layer bidirectional batch_first LSTM(128 -> 128) with biases is expected")
        w_ih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], 0).detach()              # [1024, 128]
        bias = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse], 0).detach()
        w_hh = torch.stack([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse], 0).detach()             # [2, 512, 128]
        self.w_ih_t = nn.Parameter(w_ih.t().contiguous(), requires_grad=False)                      # [128, 1024]
        self.bias = nn.Parameter(bias.clone(), requires_grad=False)
        self.w_hh = nn.Parameter(w_hh.contiguous(), requires_grad=False)

    @staticmethod
    def supports(lstm) -> bool:
        return _lstm_is_bilstm128(lstm)

    def forward(self, x):
        b, t, _ = x.shape
        gx = torch.addmm(self.bias, x.reshape(b * t, 128), self.w_ih_t).view(b, t, 2, 512)
        return _fe.bilstm128_forward(gx, self.w_hh), None


class InferenceEngine:
    """Inference-only execution of a CustomModel (the c3 path: HIP frontend + SpecAugment + CRNN forward):
      * eval-mode BatchNorm folded into the layer in front of it (`fold_batchnorm`);
      * every Conv2D + bias + ReLU (+ MaxPool) of the conv stack as MIOpen convolution + one HIP epilogue pass - except
        block 1 (1 or 2 -> 32 -> 32 channels at full resolution), whose two convolutions are HIP kernels themselves: a
        one-pass stencil and an implicit GEMM on the fp32 matrix cores with bias, ReLU and the pooling fused
        (`hip_convs=False`: MIOpen, block 1 in NCHW where its solvers are 40 % faster for the 32 -> 32 layer);
      * the bidirectional LSTM as one GEMM + ONE HIP launch for the whole recurrence (`_HipBiLSTM`);
      * frontend + forward captured into ONE hipGraph (`replay`), when a frontend and an example batch are given.
    Same function as `model.eval()(x)` up to fp32 rounding (GPU test: <= 1e-4 on the sigmoid outputs).  The model
    stays on PyTorch-ROCm (MIOpen / hipBLASLt); only the elementwise epilogues are this repository's kernels."""

    def __init__(self, model: "CustomModel", frontend: Optional["WaveFrontend"] = None,
                 example_wav: Optional[torch.Tensor] = None, fuse_epilogues: bool = True, block1_nchw: bool = True,
                 fuse_lstm: bool = True, hip_convs: bool = True):
        self.model = fold_batchnorm(model)
        self.fused_convs = 0
        self.hip_convs = 0
        self.fused_lstm = False
        dev = next(self.model.parameters()).device
        self.wino_convs = 0
        self.split_bf16_convs = 0
        if fuse_epilogues and hip_convs and SW.WINO_CONVS and dev.type == 'cuda':
            # the trailing run of plain ConvMPBlocks whose convolutions the Winograd kernel takes (blocks 2-5 of v9): one module
            feats = list(self.model.features)
            k = len(feats)
            while k > 1 and _WinoStack.eligible(feats[k - 1]):
                k -= 1
            if k < len(feats):
                stack = _WinoStack(feats[k:])
                self.wino_convs = len(stack.layers)
                self.split_bf16_convs = sum(1 for layer in stack.layers if layer[3])   # of them on the BF16 matrix cores (IRIS_WINO_SPLIT_BF16)
                self.model.features = nn.Sequential(*feats[:k], stack)
        if fuse_epilogues and dev.type == 'cuda':
            first = True
            for blk in self.model.features:
                if not isinstance(blk, ConvMPBlock):
                    continue
                nchw, first = first and block1_nchw, False
                convs = list(blk.convs)
                has_pool = isinstance(blk.pool, nn.MaxPool2d)
                ok = all(len(m) == 3 and isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.Identity) and
                         m[0].out_channels % 4 == 0 for m in convs)
                if not ok:
                    continue
                nchw = nchw and has_pool  # the hand-over to NHWC happens in the pooling epilogue
                forms = [_ConvBiasReLU.hip_form(m[0]) if hip_convs else None for m in convs]
                if any(forms):  # layers with a HIP convolution stay channels-last throughout
                    nchw = False
                blk.convs = nn.Sequential(*[_ConvBiasReLU(m[0], has_pool and i == len(convs) - 1, nchw and not forms[i], forms[i])
                                            for i, m in enumerate(convs)])
                self.hip_convs += sum(1 for f in forms if f)
                if has_pool:
                    blk.pool = nn.Identity()
                self.fused_convs += len(convs)
        self.fused_convs += self.wino_convs
        self.hip_convs += self.wino_convs
        if self.wino_convs:  # the 32 -> 32 kernel right in front of the Winograd stack writes its input layout itself
            feats = list(self.model.features)
            prev = feats[-2] if len(feats) >= 2 else None
            tail = list(prev.convs)[-1] if isinstance(prev, ConvMPBlock) and len(prev.convs) else None
            if isinstance(tail, _ConvBiasReLU) and tail.hip == 'mfma32' and isinstance(prev.pool, nn.Identity):
                tail.out_chunked = True
        if fuse_lstm and dev.type == 'cuda' and _HipBiLSTM.supports(getattr(self.model, 'lstm', None)):
            self.model.lstm = _HipBiLSTM(self.model.lstm)
            self.fused_lstm = True
        self.frontend, self.graph, self.graph_ok, self.graph_error = frontend, None, False, None
        if frontend is not None and example_wav is not None and dev.type == 'cuda':
            self.wav = example_wav
            try:
                self._capture()
                self.graph_ok = True
            except Exception as exc:  # capture is an optimisation: the eager path stays available
                self.graph_error = repr(exc)[:300]
                self.graph = None

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x)

    def eval(self):  # stands in for the model wherever one is evaluated (inference.predict_frames, metrics.evaluate's loop)
        return self

    @torch.no_grad()
    def eager(self, wav: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.model(self.frontend(self.wav if wav is None else wav))

    def _draw(self):
        fe = self.frontend
        b, n_time = self.wav.shape[0], fe.plan.num_frames(self.wav.shape[2])
        tb, fb = fe.draw_bands_device(b, n_time)
        if fe.filter_bins:
            flt = torch.tensor([[[1, fe.filter_bins]]], dtype=torch.int32, device=fe.plan.device).expand(b, 1, 2)
            fb = torch.cat([fb, flt], dim=1)
        return tb.contiguous(), fb.contiguous()

    @torch.no_grad()
    def _capture(self):
        fe, dev = self.frontend, self.frontend.plan.device
        self._tb = self._fb = None
        if fe.training:
            self._tb, self._fb = self._draw()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up: MIOpen find, geometry caches, allocator
            for _ in range(3):
                feats = fe.plan.wav_to_logmel(self.wav, minmax=fe.do_minmax, log=True, t_bands=self._tb, f_bands=self._fb)
                self.model(feats)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            feats = fe.plan.wav_to_logmel(self.wav, minmax=fe.do_minmax, log=True, t_bands=self._tb, f_bands=self._fb)
            self.out = self.model(feats)

    @torch.no_grad()
    def replay(self, wav: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Frontend (fresh SpecAugment bands, drawn on the device) + forward as one graph replay.  `wav` is copied into
        the captured input buffer; None re-uses its current contents."""
        if self.graph is None:
            return self.eager(wav)
        if wav is not None and wav.data_ptr() != self.wav.data_ptr():
            self.wav.copy_(wav)
        if self._tb is not None:
            tb, fb = self._draw()
            self._tb.copy_(tb)
            self._fb.copy_(fb)
        self.graph.replay()
        return self.out


def define_keras_model(config=None):
    """Name kept for drop-in use; returns the torch CustomModel (sj_train.py:214-255)."""
    return CustomModel(config)


def get_model(config):
    if config.model_type == 'vad':
        return define_keras_model(config)
    raise NotImplementedError(f"model_type '{config.model_type}' is outside the accelerated path "
                              "(EfficientNet / speech-enhancement branches, sj_train.py:299-401)")


# ---------------------------------------------------------------------------
# checkpoints of the reference: Keras weights -> this module's state_dict      sj_train.py:467-469, eval.py:42-65
# ---------------------------------------------------------------------------
def _keras_weight_list(weights) -> list:
    """An ORDERED list of arrays from: a list / tuple (model.get_weights()), an .npz path or an open NpzFile / dict whose
    keys are 'arr_0', 'arr_1', ... (np.savez(path, *model.get_weights())) or '<index>|<keras weight name>'
    (scripts/dump_keras_weights.py).  Order = Keras' model.weights order = layer order of define_keras_model."""
    if isinstance(weights, (str, os.PathLike)):
        with np.load(weights) as z:
            return _keras_weight_list({k: z[k] for k in z.files})
    if isinstance(weights, (list, tuple)):
        return [np.asarray(w) for w in weights]
    keys = list(weights.keys())

    def order(k):
        head = k.split('|', 1)[0]
        if head.isdigit():
            return int(head)
        if k.startswith('arr_') and k[4:].isdigit():
            return int(k[4:])
        raise ValueError(f"load_keras_weights: cannot order the key {k!r}; expected 'arr_<i>' or '<i>|<name>' keys "
                         "(np.savez(path, *model.get_weights()) or scripts/dump_keras_weights.py)")
    return [np.asarray(weights[k]) for k in sorted(keys, key=order)]


@torch.no_grad()
def load_keras_weights(model: "CustomModel", weights) -> "CustomModel":
    """Load a checkpoint of the REFERENCE model (`model.load_weights(NAME)`, sj_train.py:467-469; eval.py:42-65) into the
    torch CustomModel: `weights` = the reference model's `get_weights()` in layer order (see `_keras_weight_list`; Keras
    .h5 files are converted where TensorFlow exists by scripts/dump_keras_weights.py - h5py is not needed here).
    Layer walk of define_keras_model (sj_train.py:214-255) with Keras' layouts mapped onto torch's:
      Conv2D kernel [kh, kw, cin, cout] (HWIO) -> weight [cout, cin, kh, kw]; bias as is
      BatchNormalization gamma, beta, moving_mean, moving_variance -> weight, bias, running_mean, running_var (eps 1e-3 both)
      Dense / TimeDistributed(Dense) kernel [in, out] -> weight [out, in]; the TimeDistributed input is the Permute + Reshape
        of [B, M', T', C] to [B, T', M' C] (m' major, :243-244) - the order `CustomModel.forward` flattens in
      Bidirectional(LSTM(128)) forward then backward layer: kernel [in, 4u], recurrent_kernel [u, 4u], bias [4u], gate order
        i, f, c, o = torch's i, f, g, o -> weight_ih [4u, in], weight_hh [4u, u], bias_ih = bias, bias_hh = 0
    Shapes are checked entry by entry; a count or shape mismatch raises ValueError naming the layer.  v 6 / 7 / 8 / 9."""
    ws = _keras_weight_list(weights)
    pos = [0]

    def take(shape, what):
        if pos[0] >= len(ws):
            raise ValueError(f"load_keras_weights: ran out of arrays at {what} (got {len(ws)})")
        w = ws[pos[0]]
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"load_keras_weights: array {pos[0]} is {tuple(w.shape)}, expected {tuple(shape)} for {what}")
        pos[0] += 1
        return torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))

    def put(dst, src):
        dst.copy_(src.to(dst.device, dst.dtype))  # same shape by construction; copy_ honours dst's memory format

    def conv(c: nn.Conv2d, what):
        kh, kw = c.kernel_size
        k = take((kh, kw, c.in_channels, c.out_channels), what + ' kernel')
        put(c.weight, k.permute(3, 2, 0, 1).contiguous())
        put(c.bias, take((c.out_channels,), what + ' bias'))

    def bnorm(b, what):
        n = b.num_features
        put(b.weight, take((n,), what + ' gamma'))
        put(b.bias, take((n,), what + ' beta'))
        put(b.running_mean, take((n,), what + ' moving_mean'))
        put(b.running_var, take((n,), what + ' moving_variance'))

    def dense(fc: nn.Linear, what):
        put(fc.weight, take((fc.in_features, fc.out_features), what + ' kernel').t().contiguous())
        put(fc.bias, take((fc.out_features,), what + ' bias'))

    def conv_bn(layer: _ConvBNReLU, what):
        conv(layer[0], what)
        if isinstance(layer[1], nn.BatchNorm2d):
            bnorm(layer[1], what + ' BatchNormalization')

    for bi, blk in enumerate(model.features):
        if isinstance(blk, ConvMPBlock):
            for li, layer in enumerate(blk.convs):
                conv_bn(layer, f'features[{bi}].convs[{li}] Conv2D')
        elif isinstance(blk, _Bottleneck):
            for li, layer in enumerate(blk.body):
                conv_bn(layer, f'features[{bi}].body[{li}] Conv2D')
        # _SmoothPool has no weights
    dense(model.td, 'TimeDistributed(Dense 1024)')
    for fi, fc in enumerate(list(model.fc_pre)):
        dense(fc.fc, f'fc_pre[{fi}] Dense')
        bnorm(fc.bn, f'fc_pre[{fi}] BatchNormalization')
    if model.lstm is not None:
        u, nin = model.lstm.hidden_size, model.lstm.input_size
        for suffix, what in (('', 'Bidirectional forward LSTM'), ('_reverse', 'Bidirectional backward LSTM')):
            put(getattr(model.lstm, 'weight_ih_l0' + suffix), take((nin, 4 * u), what + ' kernel').t().contiguous())
            put(getattr(model.lstm, 'weight_hh_l0' + suffix), take((u, 4 * u), what + ' recurrent_kernel').t().contiguous())
            put(getattr(model.lstm, 'bias_ih_l0' + suffix), take((4 * u,), what + ' bias'))
            getattr(model.lstm, 'bias_hh_l0' + suffix).zero_()
    dense(model.fc_post.fc, 'fc_post Dense')
    bnorm(model.fc_post.bn, 'fc_post BatchNormalization')
    dense(model.head.fc, 'head Dense')
    if pos[0] != len(ws):
        raise ValueError(f"load_keras_weights: {len(ws) - pos[0]} arrays left over after the last layer ({len(ws)} given, "
                         f"{pos[0]} used): not a checkpoint of this architecture (v {model.config_v})")
    if hasattr(model, 'bump_generation'):
        model.bump_generation()
    return model


def keras_weight_shapes(model: "CustomModel") -> list:
    """Shapes of the reference model's get_weights() for this architecture, in order (what `load_keras_weights` expects)."""
    probe = []

    def conv_bn(layer):
        c = layer[0]
        probe.append((*c.kernel_size, c.in_channels, c.out_channels))
        probe.append((c.out_channels,))
        if isinstance(layer[1], nn.BatchNorm2d):
            probe.extend([(c.out_channels,)] * 4)
    for blk in model.features:
        if isinstance(blk, ConvMPBlock):
            for layer in blk.convs:
                conv_bn(layer)
        elif isinstance(blk, _Bottleneck):
            for layer in blk.body:
                conv_bn(layer)
    probe.extend([(model.td.in_features, model.td.out_features), (model.td.out_features,)])
    for fc in list(model.fc_pre):
        probe.extend([(fc.fc.in_features, fc.fc.out_features), (fc.fc.out_features,)] + [(fc.fc.out_features,)] * 4)
    if model.lstm is not None:
        u, nin = model.lstm.hidden_size, model.lstm.input_size
        probe.extend([(nin, 4 * u), (u, 4 * u), (4 * u,)] * 2)
    fc = model.fc_post
    probe.extend([(fc.fc.in_features, fc.fc.out_features), (fc.fc.out_features,)] + [(fc.fc.out_features,)] * 4)
    probe.extend([(model.head.fc.in_features, model.head.fc.out_features), (model.head.fc.out_features,)])
    return probe


def binary_crossentropy(y_true, y_pred):
    """tf.keras.losses.BinaryCrossentropy(): mean over all elements, probabilities
    clipped to [1e-7, 1 - 1e-7]."""
    p = torch.clamp(y_pred, 1e-7, 1 - 1e-7)
    if y_true.dtype == p.dtype and y_true.shape == p.shape:
        # the same sum as below in TWO launches forward and two backward instead of nine and ten (ATen's own clamp of log at -100
        # never acts on probabilities clipped to 1e-7): ~0.1 ms of small kernels per training step (scripts/gpu_op_census.py)
        return torch.nn.functional.binary_cross_entropy(p, y_true)
    return torch.mean(-(y_true * torch.log(p) + (1 - y_true) * torch.log(1 - p)))
