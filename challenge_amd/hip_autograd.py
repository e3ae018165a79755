"""The HIP passes of the CRNN training step as torch.autograd Functions, and their bookkeeping.

Every Function here wraps raw-pointer entry points of libiris_frontend.so (include/iris_frontend.h) around the layers of
define_keras_model (sj_train.py:191-255 upstream): training-mode bias + BatchNorm + ReLU (+ MaxPool), the first layer with its
convolution recomputed inside the passes, the Winograd / implicit-GEMM convolutions with their backward-data and weight-gradient
passes, the bidirectional LSTM's recurrence; plus AGC + clipvalue for a whole model in one launch (sj_train.py:145-155) and the
zero-initialised scratch pool the passes share.  `model.py` decides per layer which of them runs (switches.py); each one is
compared with the stock torch / MIOpen operator in tests/test_transforms_gpu.py.  No CPU fallback: CPU tensors take the stock
torch ops in model.py and never reach this module."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import frontend as _fe
from . import switches as SW
from .utils import unitwise_norm


def adaptive_clip_grad(parameters, gradients, clip_factor=0.01, eps=1e-3):
    """Adaptive gradient clipping (sj_train.py:145-155): per output unit, rescale g to
    max_norm = max(||p||, eps) * clip_factor where ||g|| >= max_norm."""
    new_grads = []
    for params, grads in zip(parameters, gradients):
        if grads is None:
            new_grads.append(None)
            continue
        p_norm = unitwise_norm(params.detach())
        max_norm = torch.clamp(p_norm, min=eps) * clip_factor
        grad_norm = unitwise_norm(grads)
        clipped = grads * (max_norm / torch.clamp(grad_norm, min=1e-6))
        new_grads.append(torch.where(grad_norm < max_norm, grads, clipped))
    return new_grads


class FusedAGC:
    """adaptive_clip_grad + clipvalue for a whole model in ONE HIP launch (iris_agc_clip): a
    device table with one record per output unit (row of a Linear/LSTM weight, output
    channel of a conv kernel, or a whole 1-D tensor).  The table is rebuilt only when a
    parameter or gradient buffer moves."""

    def __init__(self, params):
        self.params = [p for p in params]
        self._adam = None   # a torch.optim.Adam whose update rides in the same launch (`adam_step`), or None
        self._sig = None
        self._table = None
        self._slow = []
        # tables by (parameter, gradient) address set: with the gradients dropped every step (zero_grad(set_to_none=True): no
        # zero fills, and AccumulateGrad takes the incoming gradient instead of adding it to a zeroed buffer - 84 launches
        # fewer per step of the v9 CRNN, profiles/r4/accum_probe.log) the caching allocator hands the gradient buffers back
        # at a small number of recurring address sets (2 observed), each of which gets its table once
        self._cache = {}

    @staticmethod
    def _rows_of(p):
        if p.dim() <= 1:
            return 1, p.numel()
        return p.shape[0], p.numel() // p.shape[0]

    def _build(self):
        import ctypes as C
        # Layout of the table for a given classification of the parameters (which have a gradient, in which layout) is the
        # same every step - only the gradients' base addresses move when the step drops its gradients: the per-row offsets
        # are built once per classification and a step only adds this step's gradient addresses (0.7 -> 0.1 ms of host time)
        fast, self._slow = [], []
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            rows, length = self._rows_of(p)
            ok = p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_cuda
            if p.dim() > 1:
                ok = ok and p.stride(0) == length and g.stride(0) == length
                ok = ok and min(p.stride()[1:]) == 1 and min(g.stride()[1:]) == 1
            else:
                ok = ok and p.is_contiguous() and g.is_contiguous()
            (fast if ok else self._slow).append(p)
        key = tuple(id(p) for p in fast)
        plan = getattr(self, '_plan', None)
        if plan is None or plan[0] != key:
            rows_len = [self._rows_of(p) for p in fast]
            counts = np.array([r for r, _ in rows_len], np.int64)
            rep = np.repeat(np.arange(len(fast)), counts)                         # table row -> parameter
            within = np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(np.cumsum(counts) - counts, counts)
            length = np.array([l for _, l in rows_len], np.int64)[rep]
            plan = self._plan = (key, rep, within * length * 4, length)
        _, rep, offs, length = plan
        cols = self._cols()
        table = np.empty((rep.shape[0], cols), np.int64)
        if fast:
            table[:, 0] = np.array([p.data_ptr() for p in fast], np.int64)[rep] + offs
            table[:, 1] = np.array([p.grad.data_ptr() for p in fast], np.int64)[rep] + offs
            table[:, 2] = length
            if cols == 5:   # iris_agc_adam_row: the two moments lie in the parameter's own layout
                st = self._adam.state
                table[:, 3] = np.array([st[p]['exp_avg'].data_ptr() for p in fast], np.int64)[rep] + offs
                table[:, 4] = np.array([st[p]['exp_avg_sq'].data_ptr() for p in fast], np.int64)[rep] + offs
        recs = [table]
        table = np.concatenate(recs) if recs else np.zeros((0, cols), np.int64)
        # pinned staging + asynchronous copy: legal while a hipGraph is being captured (it becomes a copy node of the graph).
        # Under capture the buffers must already exist (`reserve`, called by GraphedTrainStep before the capture starts:
        # allocating pinned memory inside a capture invalidates it); the staging buffer stays alive for as long as a captured
        # copy may replay from it.
        reserved = getattr(self, '_reserved', None)
        if reserved is not None and reserved[0].shape[0] >= table.shape[0] and reserved[0].shape[1] == cols:
            # `reserve` sized the buffers for EVERY parameter having a gradient in a layout the kernel takes; a parameter
            # without a gradient or on the torch path (`_slow`) only makes the table shorter: fill a prefix and hand the
            # kernel the actual row count (no allocation inside a capture whatever the row count turns out to be)
            n = int(table.shape[0])
            host, dev_table = reserved[0][:n], reserved[1][:n]
            self._reserved = None
            host.numpy()[...] = table
            dev_table.copy_(host, non_blocking=True)
            self._host_table, self._table = host, dev_table
        elif self.params[0].is_cuda:
            # One arena of eight tables (device + pinned staging), allocated once: a table per recurring address set WITHOUT a fresh
            # allocation per build.  (Allocating each new table from torch's caching allocator moved the small-block pool the
            # gradients themselves come from: with the five-column tables their addresses never recurred, every step built and
            # pinned a new table - 1 ms, sometimes 170 ms - and the eager step read 37 ms instead of 10.)
            n, slot = int(table.shape[0]), len(self._cache) % 8
            arena = getattr(self, '_arena', None)
            rows = sum(self._rows_of(p)[0] for p in self.params)
            if arena is None or arena[0].shape[1] < rows or arena[0].shape[2] != cols:
                arena = self._arena = (torch.empty((8, rows, cols), dtype=torch.int64).pin_memory(),
                                       torch.empty((8, rows, cols), dtype=torch.int64, device=self.params[0].device))
            host, dev_table = arena[0][slot][:n], arena[1][slot][:n]
            host.numpy()[...] = table
            dev_table.copy_(host, non_blocking=True)
            self._host_table, self._table = host, dev_table
        else:
            self._table = torch.from_numpy(table)
        self._sig = self._signature()

    def _signature(self):
        """(parameter address, gradient address, gradient strides) per parameter: a gradient buffer handed back at the same
        address in another layout must not reuse a table built for the old one (fast / slow classification, row stride).
        With an optimiser attached: + the address of its first moment (a replaced state means a new table)."""
        adam = self._adam
        return tuple((p.data_ptr(),) + ((-1, ()) if p.grad is None else (p.grad.data_ptr(), tuple(p.grad.stride())))
                     + ((adam.state[p]['exp_avg'].data_ptr(),) if adam is not None and 'exp_avg' in adam.state.get(p, {}) else ())
                     for p in self.params)

    def _cols(self) -> int:
        return 5 if self._adam is not None else 3

    def _reset_tables(self, adam) -> None:
        """Another table format from here on: forget the cached tables (their arena slots are reused from the first, so nothing
        in flight may still read them)."""
        if self._cache and self.params and self.params[0].is_cuda:
            torch.cuda.synchronize(self.params[0].device)
        self._adam, self._sig, self._cache = adam, None, {}

    # ---- the optimiser's update in the same launch (round 6) -----------------------------------------------------------------
    @staticmethod
    def adam_fusable(opt) -> bool:
        """A plain torch.optim.Adam as `make_optimizer` builds it: one group, no weight decay / amsgrad / maximize, fp32 device
        parameters, step counters on the device (fused or capturable)."""
        if not SW.FUSED_ADAM or type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        if g.get('amsgrad') or g.get('weight_decay') or g.get('maximize') or g.get('differentiable'):
            return False
        if not (g.get('fused') or g.get('capturable')) or not isinstance(g['betas'][0], float) or not isinstance(g['betas'][1], float):
            return False
        lr = g['lr']
        if torch.is_tensor(lr) and not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
            return False   # (the kernel reads a device-side learning rate as ONE fp32 value)
        return all(p.is_cuda and p.dtype == torch.float32 for p in g['params'])

    def attach_adam(self, opt) -> bool:
        """Let `adam_step` run `opt`'s update: creates the optimiser state torch would create at its first step (same keys,
        dtypes and layouts: state_dict() / load_state_dict() stay interchangeable).  False: not an optimiser this kernel covers."""
        if not self.adam_fusable(opt) or [id(p) for p in opt.param_groups[0]['params']] != [id(p) for p in self.params]:
            return False
        for p in self.params:
            st = opt.state[p]
            if 'exp_avg' not in st:
                st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not (torch.is_tensor(st['step']) and st['step'].is_cuda and st['step'].dtype == torch.float32
                    and st['exp_avg'].stride() == p.stride() and st['exp_avg_sq'].stride() == p.stride()
                    and st['exp_avg'].dtype == torch.float32 and st['exp_avg_sq'].dtype == torch.float32):
                return False
        if self._adam is not opt:
            self._reset_tables(opt)
        return True

    def adam_step(self, clip_factor=0.01, eps=1e-3, clipvalue=None, use_agc=True) -> bool:
        """AGC + clipvalue + the attached optimiser's Adam update in ONE launch (iris_agc_clip_adam).  False (nothing done): a
        parameter without a gradient or with a layout the kernel does not take - the caller then runs the two steps apart."""
        import ctypes as C
        if getattr(self, '_frozen', False):
            raise RuntimeError("FusedAGC: this instance belongs to a captured hipGraph (GraphedTrainStep) and cannot be called eagerly")
        opt = self._adam
        if opt is None or any(p.grad is None for p in self.params):
            return False
        sig = self._signature()
        if sig != self._sig:
            hit = self._cache.get(sig)
            if hit is not None:
                self._sig, self._table, self._slow, self._host_table = sig, hit[0], hit[1], hit[2]
            else:
                if len(self._cache) >= 8:   # the arena's slots are reused from the first: nothing in flight may still read them
                    torch.cuda.synchronize(self.params[0].device)
                    self._cache.clear()
                self._build()
                self._cache[self._sig] = (self._table, self._slow, getattr(self, '_host_table', None))
        if self._slow or self._table.shape[1] != 5:
            return False
        g = opt.param_groups[0]
        steps = [opt.state[p]['step'] for p in self.params]
        torch._foreach_add_(steps, 1)
        lr = g['lr']
        dev = self.params[0].device
        from . import _native as N
        with torch.cuda.device(dev):
            rc = N.lib().iris_agc_clip_adam(self._table.data_ptr(), int(self._table.shape[0]), float(clip_factor), float(eps),
                                            float(clipvalue or 0.0), 1 if use_agc else 0,
                                            lr.data_ptr() if torch.is_tensor(lr) else None, 0.0 if torch.is_tensor(lr) else float(lr),
                                            float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), steps[0].data_ptr(),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        N.check(rc, "iris_agc_clip_adam")
        return True

    def reserve(self) -> None:
        """Allocate the table and its pinned staging buffer NOW (outside any capture), sized for every parameter having a
        gradient in a layout the kernel takes; the next `_build` fills them in place."""
        rows, cols = sum(self._rows_of(p)[0] for p in self.params), self._cols()
        host = torch.empty((rows, cols), dtype=torch.int64)
        if self.params[0].is_cuda:
            host = host.pin_memory()
        self._reserved = (host, torch.empty((rows, cols), dtype=torch.int64, device=self.params[0].device))

    def freeze(self) -> None:
        """After a hipGraph capture: the table and its pinned staging buffer are referenced by the graph and must never be
        rebuilt; any later call of this object raises instead."""
        self._frozen = True

    def __call__(self, clip_factor=0.01, eps=1e-3, clipvalue=None):
        import ctypes as C
        if getattr(self, '_frozen', False):
            raise RuntimeError("FusedAGC: this instance belongs to a captured hipGraph (GraphedTrainStep) and cannot be "
                               "called eagerly; eager steps use the model's own instance")
        if self._adam is not None:   # AGC alone: the three-column table of iris_agc_clip (an attached optimiser's table has five)
            self._reset_tables(None)
        sig = self._signature()
        if sig != self._sig:
            hit = self._cache.get(sig)
            if hit is not None:
                self._sig, self._table, self._slow, self._host_table = sig, hit[0], hit[1], hit[2]
            else:
                if len(self._cache) >= 8:   # the arena's slots are reused from the first: nothing in flight may still read them
                    if self.params[0].is_cuda:
                        torch.cuda.synchronize(self.params[0].device)
                    self._cache.clear()
                self._build()
                self._cache[self._sig] = (self._table, self._slow, getattr(self, '_host_table', None))
        dev = self.params[0].device
        if self._table.shape[0]:
            from . import _native as N
            with torch.cuda.device(dev):
                rc = N.lib().iris_agc_clip(self._table.data_ptr(), int(self._table.shape[0]), float(clip_factor),
                                           float(eps), float(clipvalue or 0.0),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            N.check(rc, "iris_agc_clip")
        for p in self._slow:  # odd layouts: torch path
            p.grad = adaptive_clip_grad([p], [p.grad], clip_factor, eps)[0]
            if clipvalue:
                p.grad.clamp_(-clipvalue, clipvalue)


def _is_pool_2x2_same(pool):
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return (isinstance(pool, nn.MaxPool2d) and pair(pool.kernel_size) == (2, 2) and pair(pool.stride) == (2, 2)
            and pair(pool.padding) == (0, 0) and pair(pool.dilation) == (1, 1) and pool.ceil_mode and not pool.return_indices)


class _ZeroPool:
    """Zero-initialised device scratch for the fused passes (the shifted sums of the BatchNorm passes, the first layer's
    weight-gradient copies, the identically-zero bias gradients): `take` hands out slices of one buffer per dtype and
    `begin_step` re-zeroes what the previous step used with ONE fill per buffer (three) - 55 fill launches per training step fewer
    (profiles/r5/c4_step_kernel_stats.csv).  Outside a step `take` keeps handing out untouched zeros and falls back to
    torch.zeros when the buffer is exhausted.  Buffers are only ever replaced by larger ones and the old ones kept: a captured
    hipGraph (GraphedTrainStep) replays with their addresses.  A slice stays valid until the next `begin_step` on its device;
    a bias gradient that autograd adopts from a slice is zero and stays zero.
    A hipGraph REPLAY dirties the prefix its capture took without this object seeing a single `take`: the capture records how
    far it got (`marks`) and every replay reports it (`mark_dirty`), so that the next `begin_step` clears that prefix too -
    whatever a smaller model's eager step in between left the high-water mark at (advisor finding, round 5)."""

    def __init__(self):
        self._state = {}   # (device index, dtype, kind) -> [buffer, offset (high-water mark of what may be dirty), wanted]
        self._old = []

    def take(self, n: int, dtype: torch.dtype, device: torch.device, kind: str = "scratch") -> torch.Tensor:
        """`kind`: 'scratch' - sums the kernels accumulate into; 'grad' - identically-zero gradients handed to autograd.  The two
        never share a buffer: a gradient a model still holds cannot be overwritten by another model's sums (only re-zeroed)."""
        key = (device.index, dtype, kind)
        st = self._state.get(key)
        step = -(-n // 8) * 8   # 32- / 64-byte granules: every slice 16-byte aligned
        if st is None:
            st = self._state[key] = [None, 0, 0]
        st[2] += step
        if st[0] is None or st[1] + step > st[0].numel():
            return torch.zeros(n, dtype=dtype, device=device)
        out = st[0][st[1]:st[1] + n]
        st[1] += step
        return out

    def begin_step(self, device: torch.device) -> None:
        for (index, dtype, _kind), st in self._state.items():
            if index != device.index:
                continue
            if st[0] is None or st[2] > st[0].numel():   # the last step wanted more than there is: grow (already zero)
                if st[0] is not None:
                    self._old.append(st[0])
                st[0] = torch.zeros(max(2 * st[2], 4096), dtype=dtype, device=device)
            elif st[1]:
                st[0][:st[1]].zero_()
            st[1] = st[2] = 0

    def marks(self, device: torch.device) -> dict:
        """How far each buffer of `device` has been handed out since the last `begin_step`: {key: (buffer address, offset)}.
        Taken at the end of a hipGraph capture; `mark_dirty` takes it back at every replay."""
        return {key: (st[0].data_ptr(), st[1]) for key, st in self._state.items()
                if key[0] == device.index and st[0] is not None and st[1]}

    def mark_dirty(self, marks: dict) -> None:
        """A replay of the graph whose capture returned `marks` has written into those prefixes.  A buffer that has been
        replaced by a larger one since is the graph's alone (kept in `_old`, cleared by the graph's own captured fill)."""
        for key, (ptr, offset) in marks.items():
            st = self._state.get(key)
            if st is not None and st[0] is not None and st[0].data_ptr() == ptr and st[1] < offset:
                st[1] = offset


_ZERO_POOL = _ZeroPool()
_IN_STEP = [False]   # inside train_step / a GraphedTrainStep capture: the step has called begin_step itself


# Checker hook (tests, bench.py's parity leg): while a list is installed here, the BatchNorm + ReLU (+ MaxPool) passes append what
# they read and wrote - {'kind', 'z' (None for the first layer, whose z is never stored), 'y', 'gamma', 'beta', 'mean', 'rstd',
# 'pool'} - so that a reference can take the SAME ReLU / max-pool decisions this forward took (oracle/crnn_ref.py: at the full
# c4 size two correct fp32 evaluations of the network differ in a few dozen such decisions, each worth ~1e-2 of a gradient).
_TAP = [None]


class record_activations:
    """with record_activations() as tap: model(x) -> tap = one entry per fused BatchNorm pass of that forward, in call order."""

    def __enter__(self):
        self._outer, _TAP[0] = _TAP[0], []
        return _TAP[0]

    def __exit__(self, *exc):
        _TAP[0] = self._outer
        return False


def _zeros(n: int, dtype: torch.dtype, device: torch.device, kind: str = "scratch") -> torch.Tensor:
    return _ZERO_POOL.take(int(n), dtype, device, kind) if SW.ZERO_POOL else torch.zeros(int(n), dtype=dtype, device=device)


class _FusedBiasBNReLU(torch.autograd.Function):
    """y = relu(batch_norm(z + conv_bias)) in training mode on a channels_last fp32 convolution output z (sj_train.py:191-201),
    with `pool` also the block's MaxPool2d(2, 2, ceil_mode=True) behind it (the full-size y and dy then never exist).
    The bias never touches the activation: batch normalisation subtracts the batch mean, so y does not depend on it (it
    only shifts the running mean, which iris_bn_relu_apply accounts for) and its gradient is identically zero."""

    @staticmethod
    def forward(ctx, z, conv_bias, gamma, beta, running_mean, running_var, eps, momentum, pool=False, sums0=None):
        """`sums0`: (sum z, sum z^2) per channel as the convolution's epilogue accumulated them (`_WinoConv3x3(..., stats=True)`):
        the statistics pass over z is skipped."""
        import ctypes as C
        from . import _native as N
        b, c, h, w = (int(v) for v in z.shape)
        rows = b * h * w
        dev = z.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        sums = sums0 if sums0 is not None else _zeros(N.lib().iris_bn_sums_len(c), torch.float64, dev)
        if pool:
            y = torch.empty((b, c, (h + 1) // 2, (w + 1) // 2), dtype=z.dtype, device=dev, memory_format=torch.channels_last)
        else:
            y = torch.empty_like(z)  # preserves channels_last
        save_mean = torch.empty(c, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(c, dtype=torch.float32, device=dev)
        lib = N.lib()
        tail = (sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), conv_bias.data_ptr() if conv_bias is not None else None,
                float(eps), float(momentum), running_mean.data_ptr(), running_var.data_ptr(), save_mean.data_ptr(),
                save_rstd.data_ptr(), stream)
        with torch.cuda.device(dev):
            if sums0 is None:
                N.check(lib.iris_bn_stats(z.data_ptr(), rows, c, sums.data_ptr(), stream), "iris_bn_stats")
            pool_apply = lib.iris_bn_relu_pool_apply if sums0 is None else lib.iris_bn_relu_pool_apply_sums0
            apply = lib.iris_bn_relu_apply if sums0 is None else lib.iris_bn_relu_apply_sums0
            if pool:
                N.check(pool_apply(z.data_ptr(), y.data_ptr(), b, h, w, c, *tail), "iris_bn_relu_pool_apply")
            else:
                N.check(apply(z.data_ptr(), y.data_ptr(), rows, c, *tail), "iris_bn_relu_apply")
        ctx.save_for_backward(z, gamma, beta, save_mean, save_rstd)  # y is not needed: the mask is recomputed from z
        ctx.has_bias = conv_bias is not None
        ctx.pool = bool(pool)
        ctx.mark_non_differentiable(running_mean, running_var)
        if _TAP[0] is not None:
            _TAP[0].append({'kind': 'bn', 'z': z.detach(), 'y': y.detach(), 'gamma': gamma.detach(), 'beta': beta.detach(),
                            'mean': save_mean, 'rstd': save_rstd, 'pool': bool(pool)})
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _native as N
        z, gamma, beta, save_mean, save_rstd = ctx.saved_tensors
        b, c, h, w = (int(v) for v in z.shape)
        rows = b * h * w
        dev = z.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        sums = _zeros(N.lib().iris_bn_sums_len(c), torch.float64, dev)
        dz = torch.empty_like(z)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        lib = N.lib()
        stats = (save_mean.data_ptr(), save_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr())
        with torch.cuda.device(dev):
            if ctx.pool:
                N.check(lib.iris_bn_relu_pool_bwd_reduce(z.data_ptr(), dy.data_ptr(), b, h, w, c, *stats, stream),
                        "iris_bn_relu_pool_bwd_reduce")
                N.check(lib.iris_bn_relu_pool_bwd_dx(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), b, h, w, c, *stats,
                                                     dgamma.data_ptr(), dbeta.data_ptr(), stream), "iris_bn_relu_pool_bwd_dx")
            else:
                N.check(lib.iris_bn_relu_bwd_reduce(z.data_ptr(), dy.data_ptr(), rows, c, *stats, stream), "iris_bn_relu_bwd_reduce")
                N.check(lib.iris_bn_relu_bwd_dx(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), rows, c, *stats,
                                                dgamma.data_ptr(), dbeta.data_ptr(), stream), "iris_bn_relu_bwd_dx")
        dbias = _zeros(c, torch.float32, dev, "grad") if ctx.has_bias else None
        return dz, dbias, dgamma, dbeta, None, None, None, None, None, None


class _FusedConv0BNReLU(torch.autograd.Function):
    """relu(batch_norm(conv2d(x, w) + conv_bias)) for the model's FIRST layer (1 or 2 input channels, 3x3 'same') in training
    mode, the convolution recomputed inside every pass (iris_conv0_*): its output - 32x the input - is never stored.
    x gets no gradient (it is the feature tensor); the bias gradient is identically zero (BatchNorm removes the mean)."""

    @staticmethod
    def forward(ctx, x, weight, conv_bias, gamma, beta, running_mean, running_var, eps, momentum):
        import ctypes as C
        from . import _native as N
        b, cin, h, w = (int(v) for v in x.shape)
        cout = int(weight.shape[0])
        dev = x.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        xc = x.contiguous()                     # [B, CIN, H, W]; one channel: the channels_last tensor already is
        wc = weight.detach().contiguous()       # [COUT, CIN, 3, 3]
        lib = N.lib()
        sums = _zeros(lib.iris_bn_sums_len(cout), torch.float64, dev)
        y = torch.empty((b, cout, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        save_mean = torch.empty(cout, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(cout, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.iris_conv0_stats(xc.data_ptr(), wc.data_ptr(), b, cin, cout, h, w, sums.data_ptr(), stream), "iris_conv0_stats")
            N.check(lib.iris_conv0_bn_relu(xc.data_ptr(), wc.data_ptr(), y.data_ptr(), b, cin, cout, h, w, sums.data_ptr(),
                                           gamma.data_ptr(), beta.data_ptr(),
                                           conv_bias.data_ptr() if conv_bias is not None else None, float(eps), float(momentum),
                                           running_mean.data_ptr(), running_var.data_ptr(), save_mean.data_ptr(),
                                           save_rstd.data_ptr(), stream), "iris_conv0_bn_relu")
        ctx.save_for_backward(xc, wc, gamma, beta, save_mean, save_rstd)
        ctx.has_bias = conv_bias is not None
        ctx.weight_format = (torch.channels_last if weight.is_contiguous(memory_format=torch.channels_last)
                             and not weight.is_contiguous() else torch.contiguous_format)
        ctx.weight_strides = tuple(weight.stride())
        ctx.mark_non_differentiable(running_mean, running_var)
        if _TAP[0] is not None:
            _TAP[0].append({'kind': 'conv0', 'z': None, 'y': y.detach(), 'gamma': gamma.detach(), 'beta': beta.detach(),
                            'mean': save_mean, 'rstd': save_rstd, 'pool': False})
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from . import _native as N
        xc, wc, gamma, beta, save_mean, save_rstd = ctx.saved_tensors
        b, cin, h, w = (int(v) for v in xc.shape)
        cout = int(wc.shape[0])
        dev = xc.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        lib = N.lib()
        sums = _zeros(lib.iris_bn_sums_len(cout), torch.float64, dev)
        dw64 = _zeros(lib.iris_conv0_dweight_len(cin, cout), torch.float64, dev)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dev)
        dbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.iris_conv0_bn_relu_backward(xc.data_ptr(), wc.data_ptr(), dy.data_ptr(), b, cin, cout, h, w,
                                                    save_mean.data_ptr(), save_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                    sums.data_ptr(), dw64.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), stream),
                    "iris_conv0_bn_relu_backward")
        dw = dw64.view(-1, cout, cin, 3, 3).sum(0).to(torch.float32).contiguous(memory_format=ctx.weight_format)
        if cin == 1 and tuple(dw.stride()) != ctx.weight_strides:
            # one input channel: both memory formats are the same bytes, only the stride of the size-1 axis differs - hand the
            # gradient back with the parameter's own strides (DDP's bucket views follow those, and warn otherwise)
            dw = dw.as_strided(dw.shape, ctx.weight_strides)
        dbias = _zeros(cout, torch.float32, dev, "grad") if ctx.has_bias else None
        return None, dw, dbias, dgamma, dbeta, None, None, None, None


def _is_first_layer_conv(conv, x) -> bool:
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    co = conv.out_channels
    return (conv.in_channels in (1, 2) and pair(conv.kernel_size) == (3, 3) and pair(conv.padding) == (1, 1)
            and pair(conv.stride) == (1, 1) and pair(conv.dilation) == (1, 1) and conv.groups == 1
            and co % 4 == 0 and co <= 256 and 1024 % co == 0 and not x.requires_grad and x.dim() == 4
            and x.shape[3] <= 2048)


class _PackBook:
    """The Winograd weight packings of a training step, made in ONE launch at the top of the forward pass (switch FUSED_PACK).

    `_WinoConv3x3` asks `take(weight, transposed, b3)` for every packing it needs.  The first time a (weight storage, pass, kind) is
    asked for it is packed on the spot, as before, into a buffer this book keeps - and remembered; from the next forward pass on,
    `prepack()` (called by CustomModel.forward in training mode) packs everything remembered in one launch per kind, and `take`
    hands those buffers out.  A packing is used only while the weight's autograd version is the one it was packed at (any in-place
    update - the optimiser step, load_state_dict, SWA - makes it stale and `take` packs again), so a layer called on its own, two
    forwards before one backward, or an evaluation in between all stay correct.  Entries are keyed by the weight's storage address,
    shape and strides, not by the tensor object: GraphedTrainStep runs the model on ALIASES of the parameters over the same storage
    (they share the version counter).  The buffers persist, so a captured graph replays the one-launch packing and the
    convolutions on fixed addresses."""

    def __init__(self):
        self.entries = {}   # key -> [packed tensor, version it was packed at, weakref to a tensor over the weight]

    @staticmethod
    def _key(weight, transposed, b3):
        return (weight.device.index, weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()), bool(transposed), bool(b3))

    def take(self, weight, transposed: bool, b3: bool):
        if not SW.FUSED_PACK:
            return _fe.wino_pack_weights_device(weight, transposed=transposed, split_bf16=b3)
        import weakref
        key = self._key(weight, transposed, b3)
        ent = self.entries.get(key)
        # (the tensor the packing was made from must still be alive: a freed weight's address - and version 0 - comes back with
        # the next tensor of its size)
        if ent is not None and ent[1] == weight._version and ent[2]() is not None:
            return ent[0]
        co, ci = int(weight.shape[0]), int(weight.shape[1])
        cin, cout = (co, ci) if transposed else (ci, co)
        out = ent[0] if ent is not None else torch.empty(_fe.wino_packed_len(cin, cout, b3), dtype=torch.float32, device=weight.device)
        _fe.wino_pack_weights_device(weight, transposed=transposed, out=out, split_bf16=b3)
        if ent is None and len(self.entries) >= 64:   # layers called on their own with ever new weights: drop what has died
            for k in [k for k, e in self.entries.items() if e[2]() is None]:
                del self.entries[k]
        self.entries[key] = [out, weight._version, weakref.ref(weight)]
        return out

    def prepack(self, device):
        """Pack every remembered weight of `device` whose packing is stale, one launch per kind; forget weights that are gone."""
        if not SW.FUSED_PACK or not self.entries:
            return
        jobs = {False: [], True: []}
        dead = []
        for key, ent in self.entries.items():
            if key[0] != device.index:
                continue
            w = ent[2]()
            if w is None or w.data_ptr() != key[1] or tuple(w.stride()) != key[3]:
                dead.append(key)
                continue
            if ent[1] != w._version:
                jobs[key[5]].append((w, key[4], ent[0], ent))
        for key in dead:
            del self.entries[key]
        for b3, lst in jobs.items():
            if lst:
                _fe.wino_pack_weights_device_multi([(w, t, out) for w, t, out, _ in lst], split_bf16=b3)
                for w, _, _, ent in lst:
                    ent[1] = w._version

    def invalidate(self):
        """Every packing is stale (CustomModel.bump_generation: a hipGraph replay, an all-reduce into the parameters or any other
        update that does not move the weights' autograd version); the buffers are kept."""
        for ent in self.entries.values():
            ent[1] = -1

    def clear(self):
        self.entries.clear()


_PACKS = _PackBook()


class _WinoConv3x3(torch.autograd.Function):
    """z = conv2d(x, weight, padding=1) for channels_last fp32 tensors.  forward (`fwd`): iris_conv3x3_wino on the weights packed
    on the device this step, else MIOpen; backward: dx (`bwd`) by the same kernel on the transposed / flipped weights, else
    MIOpen; dW (`wrw`) by iris_conv3x3_wino_wrw, else MIOpen's weight-gradient kernel (aten.convolution_backward).
    `fwd` / `bwd` == 'c32': the 32 -> 32 layer of block 1 - forward and backward-data by the implicit-GEMM kernel of the
    inference engine without its bias / ReLU (iris_conv3x3_c32; the backward pass reads the weight transposed and flipped)."""

    @staticmethod
    def forward(ctx, x, weight, fwd=True, bwd=True, wrw=False, stats=False):
        """`stats` (only where a HIP kernel runs the forward): also return the per-channel (sum z, sum z^2) of the BatchNorm behind
        the convolution, accumulated in the kernel's epilogue - `_FusedBiasBNReLU(..., sums0=...)` then skips its pass over z."""
        from . import _native as N
        sums = None
        if stats and fwd:
            sums = _zeros(N.lib().iris_bn_sums_len(int(weight.shape[0])), torch.float64, x.device)
        if fwd == 'c32':
            z = _fe.conv3x3_c32(x, weight, bn_sums=sums)
        elif fwd:
            b3 = SW.WINO_SPLIT_BF16 and int(weight.shape[1]) % 16 == 0   # GEMMs on the BF16 matrix cores, three-term split
            z = _fe.conv3x3_wino(x, _PACKS.take(weight, False, b3), None, int(weight.shape[0]), out_nhwc=True,
                                 relu=False, split_bf16=b3, bn_sums=sums)
        else:
            z = torch.nn.functional.conv2d(x, weight, None, 1, 1)
        ctx.save_for_backward(x, weight)
        ctx.wino_bwd = bwd if bwd == 'c32' else bool(bwd)
        ctx.wino_wrw = bool(wrw)
        # (the statistics output is not differentiable: without this autograd materialises a zero "gradient" for it in every
        # backward - 13 fp64 fills per training step, found in the step's kernel trace)
        ctx.set_materialize_grads(False)
        if not stats:
            return z
        if sums is None:
            return z, None
        ctx.mark_non_differentiable(sums)
        return z, sums

    @staticmethod
    def backward(ctx, dz, _dsums=None):
        if dz is None:   # (nothing flows back through z: grads are no longer materialised, see forward)
            return None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        cin = int(weight.shape[1])
        if not dz.is_contiguous(memory_format=torch.channels_last):
            dz = dz.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        wino_dx = ctx.needs_input_grad[0] and ctx.wino_bwd
        if wino_dx and ctx.wino_bwd == 'c32':
            dx = _fe.conv3x3_c32(dz, weight, transposed=True)
        elif wino_dx:
            b3 = SW.WINO_SPLIT_BF16 and int(weight.shape[0]) % 16 == 0
            dx = _fe.conv3x3_wino(dz, _PACKS.take(weight, True, b3), None, cin, out_nhwc=True,
                                  relu=False, split_bf16=b3)
        wino_dw = ctx.needs_input_grad[1] and ctx.wino_wrw
        if wino_dw:
            dw = _fe.conv3x3_wino_wrw(x, dz, like=weight)
        need = [ctx.needs_input_grad[0] and not wino_dx, ctx.needs_input_grad[1] and not wino_dw, False]
        if need[0] or need[1]:
            gi, gw, _ = torch.ops.aten.convolution_backward(dz, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, need)
            dx = gi if need[0] else dx
            dw = gw if need[1] else dw
        return dx, dw, None, None, None, None


def _wino_train_conv(conv: nn.Conv2d, x: torch.Tensor):
    """(forward by Winograd?, backward-data by Winograd?, weight gradient by Winograd?) for this layer and input, or None:
    MIOpen for everything."""
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    if not (SW.WINO_TRAIN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and pair(conv.kernel_size) == (3, 3) and pair(conv.padding) == (1, 1) and pair(conv.stride) == (1, 1)
            and pair(conv.dilation) == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() < (1 << 30)
            and x.shape[0] * x.shape[2] * x.shape[3] * max(conv.in_channels, conv.out_channels) < (1 << 30)):
        return None
    ci, co, big = conv.in_channels, conv.out_channels, max(conv.in_channels, conv.out_channels)
    fwd = ci % 8 == 0 and co % 64 == 0 and big >= SW.WINO_TRAIN_MIN_C_FWD
    bwd = co % 8 == 0 and ci % 64 == 0 and big >= SW.WINO_TRAIN_MIN_C_BWD
    wrw = SW.WINO_TRAIN_WRW and ci % 32 == 0 and co % 32 == 0 and x.shape[0] * x.shape[2] * x.shape[3] * big < (1 << 29)
    if SW.C32_TRAIN and ci == 32 and co == 32:   # block 1's second layer: the inference engine's fp32-MFMA kernel, bare
        fwd = bwd = 'c32'
    return (fwd, bwd, wrw) if (fwd or bwd or wrw) else None


class _BiLSTM128(torch.autograd.Function):
    """out = recurrence(gx, w_hh) of a bidirectional LSTM(128) with both passes through time inside ONE HIP launch each
    (iris_bilstm128_forward / _backward).  backward returns dgx (autograd carries it on into W_ih, the biases and x through
    the GEMM that formed gx) and dW_hh[d] = dgx[:, :, d, :]^T . h_prev, h_prev = the output shifted by one step of d."""

    @staticmethod
    def forward(ctx, gx, w_hh):
        out, act = _fe.bilstm128_forward(gx, w_hh, save=True)
        ctx.save_for_backward(act, w_hh, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        act, w_hh, out = ctx.saved_tensors
        dgx = _fe.bilstm128_backward(dout, act, w_hh)
        b, t = out.shape[0], out.shape[1]
        hprev = torch.zeros((2, b, t, 128), dtype=out.dtype, device=out.device)
        if t > 1:
            hprev[0, :, 1:] = out[:, :-1, :128]   # forward direction came from t - 1
            hprev[1, :, :-1] = out[:, 1:, 128:]   # backward direction came from t + 1
        dg = dgx.permute(2, 3, 0, 1).reshape(2, 512, b * t)            # [d, gate row, (b, t)]
        dw_hh = torch.bmm(dg, hprev.reshape(2, b * t, 128))
        return dgx, dw_hh


def _lstm_is_bilstm128(lstm) -> bool:
    return (isinstance(lstm, nn.LSTM) and lstm.input_size == 128 and lstm.hidden_size == 128 and lstm.num_layers == 1
            and lstm.bidirectional and lstm.batch_first and lstm.bias and lstm.proj_size == 0
            and lstm.weight_ih_l0.dtype == torch.float32)


def bilstm128(lstm: nn.LSTM, x: torch.Tensor) -> torch.Tensor:
    """`lstm(x)[0]` for the model's nn.LSTM(128, 128, bidirectional, batch_first) with the recurrence - and, under
    autograd, its back-propagation through time - in one HIP launch each; the parameters stay the module's own."""
    b, t, _ = x.shape
    w_ih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], 0)                               # [1024, 128]
    bias = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse], 0)
    w_hh = torch.stack([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse], 0)                              # [2, 512, 128]
    gx = torch.nn.functional.linear(x.reshape(b * t, 128), w_ih, bias).view(b, t, 2, 512)
    if torch.is_grad_enabled() and (gx.requires_grad or w_hh.requires_grad):
        return _BiLSTM128.apply(gx, w_hh)
    return _fe.bilstm128_forward(gx, w_hh)
