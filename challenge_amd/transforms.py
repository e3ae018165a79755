"""Drop-in counterpart of the reference's transforms.py on torch tensors.

Same callable names, arguments, defaults and `(x, y=None) -> x | (x, y)` map
convention as the reference (transforms.py:12-195).  The functions on the live
hot path -- `mask`, `complex_to_magphase`, `magphase_to_mel` -- run as HIP kernels
through the C ABI and therefore need tensors on a ROCm device (no CPU fallback).
The remaining signatures (`random_shift`, `log_magphase`, `minmax_norm_magphase`,
`magphase_to_complex`, `phase_vocoder`; only the reference's tests call them) are
device-agnostic torch glue.

Randomness: the reference draws from TensorFlow's global Philox stream, which
cannot be reproduced.  Every random transform here is *draw* (documented
distribution, NumPy Generator, `set_seed`) + deterministic *apply*; the apply
halves are what the parity tests pin (`mask_apply`, `random_shift_apply`)."""
from __future__ import annotations

from math import e, log
from typing import Optional

import numpy as np
import torch

from . import frontend as _fe
from .dataset import AUTOTUNE  # noqa: F401  (re-exported like transforms.py:6)

EPSILON = 1e-8
LOG_EPSILON = log(EPSILON) / log(e)

_rng = np.random.default_rng()


def set_seed(seed: Optional[int]) -> None:
    """Seed the host RNG used by the random transforms (tf.random.set_seed analogue)."""
    global _rng
    _rng = np.random.default_rng(seed)


def get_rng() -> np.random.Generator:
    return _rng


# ---------------------------------------------------------------------------
# FEATURE INDEPENDENT AUGMENTATIONS
# ---------------------------------------------------------------------------
def mask_draw(total: int, max_mask_size: Optional[int] = None, n_mask: int = 1,
              rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """int32 [n_mask, 2] of (offset, size): size ~ U{0..max_mask_size-1},
    offset ~ U{0..total-size-1} (transforms.py:25-26)."""
    rng = _rng if rng is None else rng
    if max_mask_size is None:
        max_mask_size = total
    if max_mask_size <= 0:
        raise ValueError("mask: max_mask_size must be positive")
    bands = np.zeros((n_mask, 2), np.int32)
    for i in range(n_mask):
        size = int(rng.integers(0, max_mask_size))
        if total - size <= 0:
            raise ValueError("mask: maxval must be > 0 (mask of size %d on an axis of %d)" % (size, total))
        bands[i] = (int(rng.integers(0, total - size)), size)
    return bands


def mask_draw_batch(batch: int, total: int, max_mask_size: Optional[int] = None, n_mask: int = 1,
                    rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """`mask_draw` for a whole batch in two vectorised draws: int32 [batch, n_mask, 2] of (offset, size), every
    (sample, mask) pair independent with the distributions of transforms.py:25-26."""
    rng = _rng if rng is None else rng
    if max_mask_size is None:
        max_mask_size = total
    if max_mask_size <= 0:
        raise ValueError("mask: max_mask_size must be positive")
    size = rng.integers(0, max_mask_size, size=(batch, n_mask))
    if batch * n_mask and int(size.max()) >= total:
        raise ValueError("mask: maxval must be > 0 (mask of size %d on an axis of %d)" % (int(size.max()), total))
    off = rng.integers(0, total - size)  # array `high`: one independent draw per element
    return np.stack([off, size], axis=-1).astype(np.int32)


def mask_apply(specs: torch.Tensor, axis: int, bands) -> torch.Tensor:
    """specs with the bands [offset, offset+size) zeroed along `axis`, in specs.dtype."""
    return _fe.mask_apply(specs, axis, bands)


def mask(specs: torch.Tensor, axis: int, max_mask_size: Optional[int] = None, n_mask: int = 1) -> torch.Tensor:
    """SpecAugment band mask (transforms.py:12-40): multiply by `n_mask` random 0/1 bands."""
    total = specs.shape[axis]
    return mask_apply(specs, axis, mask_draw(total, max_mask_size, n_mask))


def random_shift_apply(specs: torch.Tensor, axis: int, width: int, offset: int) -> torch.Tensor:
    axis = axis % specs.dim()
    pad = [0, 0] * specs.dim()
    pad[2 * (specs.dim() - 1 - axis)] = width
    pad[2 * (specs.dim() - 1 - axis) + 1] = width
    padded = torch.nn.functional.pad(specs, pad)
    return padded.narrow(axis, offset, specs.shape[axis]).contiguous()


def random_shift(specs: torch.Tensor, axis: int = 0, width: int = 16) -> torch.Tensor:
    """Zero-pad `width` both sides of `axis`, crop the original extent at a uniform
    offset in [0, 2*width] (transforms.py:43-47)."""
    return random_shift_apply(specs, axis, width, int(_rng.integers(0, 2 * width + 1)))


# ---------------------------------------------------------------------------
# MAGNITUDE-PHASE SPECTROGRAM
# ---------------------------------------------------------------------------
def magphase_to_mel(num_mel_bins: int = 80, num_spectrogram_bins: int = 257, sample_rate: float = 16000,
                    **kwargs):
    """Closure factory (transforms.py:51-77).  The weight matrix is built once here
    (argument errors surface at creation, as in the reference); the closure maps
    [B, F, T, 2C] -> [B, M, T, C] or [F, T, 2C] -> [M, T, C].

    Extension: `mel_matrix=W` ([F, M] float32) replaces the built-in recipe - pass the matrix your TensorFlow build
    returns from tf.signal.linear_to_mel_weight_matrix for bit-exact weights."""
    external = kwargs.pop("mel_matrix", None)
    unknown = set(kwargs) - {"lower_edge_hertz", "upper_edge_hertz"}
    if unknown:
        raise TypeError(f"unexpected keyword arguments {sorted(unknown)}")
    if external is not None:
        mel_matrix = np.ascontiguousarray(external, np.float32)
        if mel_matrix.shape != (num_spectrogram_bins, num_mel_bins):
            raise ValueError(f"mel_matrix must be [{num_spectrogram_bins}, {num_mel_bins}], got {mel_matrix.shape}")
    else:
        mel_matrix = _fe.mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate, **kwargs)
    n_fft = 2 * (num_spectrogram_bins - 1)
    fft_ok = n_fft in (256, 512, 1024, 2048)
    plans = {}

    def _plan(device: torch.device, chan: int, batch: int) -> "_fe.FrontendPlan":
        key = (device.index, chan)
        plan = plans.get(key)
        if plan is None or plan.max_batch < batch:
            cap = max(batch, 2 * plan.max_batch if plan else 1)
            if fft_ok:
                plan = _fe.FrontendPlan(n_fft, None, num_mel_bins, sample_rate, chan, cap, n_fft, device,
                                        mel_matrix=mel_matrix)
            else:  # mel-only plan for an arbitrary bin count
                plan = _fe.FrontendPlan.mel_only(num_mel_bins, num_spectrogram_bins, chan, cap, device, mel_matrix)
            plans[key] = plan
        return plan

    def _magphase_to_mel(x, y=None):
        if x.dim() not in (3, 4):
            raise ValueError("len(x.shape) must be 3 or 4")
        xb = x if x.dim() == 4 else x.unsqueeze(0)
        if xb.shape[1] != num_spectrogram_bins:
            raise ValueError(f"expected {num_spectrogram_bins} spectrogram bins on axis -3, got {xb.shape[1]}")
        chan = xb.shape[-1] // 2
        if not xb.is_cuda:
            raise RuntimeError("magphase_to_mel runs as a HIP kernel: x must be on a ROCm device (no CPU fallback)")
        plan = _plan(xb.device, chan, xb.shape[0])
        mel = plan.magmel(xb.float(), is_magphase=True)  # phase half is ignored (transforms.py:64)
        if x.dim() == 3:
            mel = mel[0]
        if y is None:
            return mel
        return mel, y

    _magphase_to_mel.mel_matrix = mel_matrix
    return _magphase_to_mel


def log_magphase(specs: torch.Tensor, labels=None, n_chan: int = 2):
    """ln(x + EPSILON) on the first n_chan trailing channels, the rest passes through
    (transforms.py:80-86)."""
    if not torch.is_floating_point(specs):
        specs = specs.to(torch.float32)
    specs = torch.cat([torch.log(specs[..., :n_chan] + EPSILON), specs[..., n_chan:]], dim=-1)
    if labels is not None:
        return specs, labels
    return specs


def minmax_norm_magphase(specs: torch.Tensor, labels=None):
    """(x - min) / (max - min + EPSILON) per sample, separately for the magnitude and
    the phase halves (transforms.py:89-107)."""
    n_chan = specs.shape[-1] // 2
    axis = tuple(range(1, specs.dim()))
    out = []
    for part in (specs[..., :n_chan], specs[..., n_chan:]):
        mx = torch.amax(part, dim=axis, keepdim=True)
        mn = torch.amin(part, dim=axis, keepdim=True)
        out.append((part - mn) / (mx - mn + EPSILON))
    specs = torch.cat(out, dim=-1)
    if labels is not None:
        return specs, labels
    return specs


# ---------------------------------------------------------------------------
# COMPLEX-SPECTROGRAMS
# ---------------------------------------------------------------------------
def complex_to_magphase(complex_tensor: torch.Tensor, y=None):
    """[..., 2C] (re block, im block) -> (|z|, atan2(im, re)) (transforms.py:111-123)."""
    magphase = _fe.complex_to_magphase(complex_tensor)
    if y is None:
        return magphase
    return magphase, y


def magphase_to_complex(magphase: torch.Tensor) -> torch.Tensor:
    """Inverse of complex_to_magphase (transforms.py:126-134)."""
    if magphase.is_cuda:
        return _fe.magphase_to_complex(magphase)
    n_chan = magphase.shape[-1] // 2
    mag, phase = magphase[..., :n_chan], magphase[..., n_chan:]
    return torch.cat([mag * torch.cos(phase), mag * torch.sin(phase)], dim=-1)


def phase_vocoder(complex_spec: torch.Tensor, rate: float = 1.0) -> torch.Tensor:
    """Time-stretch a [freq, time, chan*2] spectrogram by `rate` (transforms.py:137-195):
    hop_length = freq - 1, phase advance linspace(0, pi*hop, freq), wrapped phase
    differences accumulated with the first frame's phase prepended, magnitudes linearly
    interpolated.  Output time length ceil(time / rate)."""
    if rate == 1:
        return complex_spec
    spec = complex_spec
    freq = spec.shape[0]
    hop_length = freq - 1
    n_chan = spec.shape[-1] // 2
    dt = spec.dtype

    def angle(s):
        return torch.atan2(s[..., n_chan:], s[..., :n_chan])

    phase_advance = torch.linspace(0.0, float(np.pi * hop_length), freq, dtype=dt, device=spec.device).reshape(-1, 1, 1)
    time_steps = torch.arange(0, spec.shape[1], rate, dtype=dt, device=spec.device)
    padded = torch.nn.functional.pad(spec, (0, 0, 0, 2))
    i0 = time_steps.to(torch.int64)
    i1 = (time_steps + 1).to(torch.int64)
    spec_0, spec_1 = padded[:, i0], padded[:, i1]
    angle_0, angle_1 = angle(spec_0), angle(spec_1)
    norm_0 = torch.sqrt(spec_0[..., :n_chan] ** 2 + spec_0[..., n_chan:] ** 2)
    norm_1 = torch.sqrt(spec_1[..., :n_chan] ** 2 + spec_1[..., n_chan:] ** 2)
    phase_0 = angle(padded[:, :1])
    phase = angle_1 - angle_0 - phase_advance
    phase = phase - 2 * np.pi * torch.round(phase / (2 * np.pi))  # round half to even, as tf.math.round
    phase = phase + phase_advance
    phase = torch.cat([phase_0, phase[:, :-1]], dim=1)
    phase_acc = torch.cumsum(phase, dim=1)
    alphas = (time_steps % 1.0).reshape(1, -1, 1)
    mag = alphas * norm_1 + (1 - alphas) * norm_0
    return torch.cat([mag * torch.cos(phase_acc), mag * torch.sin(phase_acc)], dim=-1)
