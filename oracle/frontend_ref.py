"""CPU oracle for the audio feature frontend -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a NumPy restatement of the reference's hot path

    waveform -> normalize -> STFT -> magnitude -> mel filterbank -> min-max -> log
    (+ SpecAugment band masks, channel helpers, mixing)

written from the behaviour of the reference (file:line citations are into
/root/reference, the read-only upstream tree), not copied from it.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path (``challenge_amd``) never does:
it calls the HIP library through the C ABI and fails loudly if that is missing.

Pinning status
--------------
* STFT: pinned.  ``torchaudio.transforms.Spectrogram(n_fft, power=None)``
  (data_utils.py:17) executes ``torch.stft(center=True, pad_mode='reflect',
  window=hann_window(n_fft), hop=n_fft//2, normalized=False, onesided=True)``;
  ``tests/golden/make_golden.py`` ran exactly that call in the build container
  (torch 2.10 CPU) and the outputs are committed under ``tests/golden``;
  ``tests/test_oracle.py`` checks this restatement against them.
* magnitude / log / min-max / mask-apply / shift-apply / phasors: pinned by the
  known-answer vectors in the reference's own tests (transforms_test.py:10-108),
  restated as data in ``tests/golden/ref_kats.json``.
* mel weight matrix: PARITY UNPINNED.  The matrix comes from
  ``tf.signal.linear_to_mel_weight_matrix`` (tensorflow-gpu==2.2.0,
  requirements.txt:1; call site transforms.py:55-56).  TensorFlow is neither in
  /root/reference nor installable here, and the reference's test checks only the
  output *shape* (transforms_test.py:45-55).  ``linear_to_mel_weight_matrix``
  below restates TF's published fp32 recipe (HTK mel, 1127*ln(1+f/700),
  triangles linear in mel, DC bin zeroed); it is cross-checked against an
  independent fp64 evaluation of the same formula (max abs difference ~1e-5:
  fp32 rounding of mel values near 2000 divided by ~24-mel-wide bands), but not
  against TF output.  If TF evaluated the recipe in fp64 and cast, W would move
  by up to that much; the stated mel tolerance is therefore relative to THIS
  matrix, which is uploaded to the GPU verbatim (never recomputed on device).
* RNG streams (mask sizes/offsets, mixing gains/crops): the reference draws from
  TF's Philox stream, which cannot be reproduced without TF.  Every random
  function is therefore split into *draw* (documented distribution) and
  deterministic *apply*; only *apply* is parity-checked.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

EPSILON = 1e-8  # transforms.py:7, utils.py:6
LOG_EPSILON = math.log(EPSILON)  # transforms.py:8

_MEL_BREAK_FREQUENCY_HERTZ = 700.0
_MEL_HIGH_FREQUENCY_Q = 1127.0


# --------------------------------------------------------------------------
# R1: load_wav tail -- normalize + STFT + layout            data_utils.py:9-34
# --------------------------------------------------------------------------
def normalize(wav: np.ndarray) -> np.ndarray:
    """wav / (10 * rms), rms over *all* channels jointly (data_utils.py:32-34)."""
    wav = np.asarray(wav, dtype=np.float32)
    rms = np.sqrt(np.mean(np.square(wav), dtype=np.float32), dtype=np.float32) * np.float32(10)
    return (wav / rms).astype(np.float32)


def hann_periodic(n: int, dtype=np.float32) -> np.ndarray:
    """torch.hann_window(n) (periodic): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def n_frames(length: int, hop: int) -> int:
    """center=True framing: T = 1 + L // hop."""
    return 1 + length // hop


def reflect_index(i: np.ndarray, length: int) -> np.ndarray:
    """Index map of torch 'reflect' padding (edge sample not repeated)."""
    i = np.abs(i)
    return np.where(i >= length, 2 * (length - 1) - i, i)


def frame_signal(wav: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """[..., L] -> [..., T, n_fft] frames of the reflect-padded signal.

    Frame t covers padded samples [t*hop, t*hop + n_fft), i.e. original samples
    t*hop - n_fft/2 + n (torch.stft center=True, pad_mode='reflect').
    """
    length = wav.shape[-1]
    if n_fft // 2 >= length:
        raise ValueError("reflect padding needs n_fft//2 < signal length")
    t = n_frames(length, hop)
    # reflect-pad once (1-D index map), then strided windows: the same samples as wav[..., reflect_index(t*hop -
    # n_fft/2 + n)], without a [T, n_fft] fancy index (whose result NumPy lays out index-major, i.e. strided)
    padded = wav[..., reflect_index(np.arange(-(n_fft // 2), length + n_fft // 2), length)]
    win = np.lib.stride_tricks.sliding_window_view(padded, n_fft, axis=-1)[..., ::hop, :]
    return win[..., :t, :]


def stft(wav: np.ndarray, n_fft: int = 512, hop: Optional[int] = None,
         dtype=np.float32) -> np.ndarray:
    """Spectrogram(n_fft, power=None) on wav[C, L] -> complex [C, F, T].

    win_length = n_fft, hop = n_fft // 2 unless given, periodic Hann, centre +
    reflect pad, one-sided, no normalisation (data_utils.py:17, :23).
    ``dtype=np.float64`` evaluates the same definition in double precision
    (used by tests to size the fp32 tolerance).
    """
    hop = n_fft // 2 if hop is None else hop
    wav = np.asarray(wav, dtype=dtype)
    frames = frame_signal(wav, n_fft, hop) * hann_periodic(n_fft, dtype)
    spec = np.fft.rfft(frames, n=n_fft, axis=-1)  # [..., T, F]
    ctype = np.complex64 if dtype == np.float32 else np.complex128
    return np.swapaxes(spec, -1, -2).astype(ctype)  # [..., F, T]


def to_ref_layout(spec: np.ndarray) -> np.ndarray:
    """complex [C, F, T] -> real [F, T, 2C], re block then im block.

    data_utils.py:26-27: [chan,freq,time,2] -> transpose(1,2,3,0) = [F,T,2,C]
    -> reshape [F, T, 2C].  Batched input [B, C, F, T] -> [B, F, T, 2C].
    """
    re = np.moveaxis(spec.real, -3, -1)
    im = np.moveaxis(spec.imag, -3, -1)
    return np.concatenate([re, im], axis=-1).astype(
        np.float32 if spec.dtype == np.complex64 else np.float64)


def resample_taps(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """The polyphase filter of torchaudio.functional.resample, resampling_method "sinc_interp_hann" - what
    torchaudio.compliance.kaldi.resample_waveform (data_utils.py:20-21) computes with.  torchaudio is a third-party dependency of
    the reference (requirements.txt:5, UNPINNED, source not under /root/reference) and is not installed here, so this restates
    its published algorithm and is pinned only by the independent evaluation of the same formula in tests/test_oracle.py:
    PARITY UNPINNED against torchaudio's own outputs.  Returns (taps [new, K] float64, width, orig, new) on the reduced ratio:
        f = rolloff min(o, n);  w = ceil(lowpass_filter_width o / f);  K = 2 w + o
        t[j, k] = clamp(f ((k - w) / o - j / n), -lpw, lpw)
        taps[j, k] = (f / o) sinc(pi t) cos^2(pi t / (2 lpw))"""
    if int(orig_freq) != orig_freq or int(new_freq) != new_freq or orig_freq <= 0 or new_freq <= 0:
        raise ValueError("sample rates must be positive integers")
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    t = np.arange(0, -n, -1, dtype=np.float64)[:, None] / n + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * np.pi / lowpass_filter_width / 2) ** 2
    t = t * np.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t))
    return sinc * window * (base / o), width, o, n


def resample_waveform(wav: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """kaldi.resample_waveform(wav, orig_freq, new_freq) of data_utils.py:20-21 for wav [C, L] (see `resample_taps`): the
    waveform padded with `width` zeros in front and `width + o` behind, a strided correlation with the n phase filters (conv1d,
    stride o), phases interleaved, cut to ceil(n L / o) samples.  fp64 sums; the result in wav's dtype."""
    wav = np.asarray(wav)
    if orig_freq == new_freq:
        return wav
    taps, width, o, n = resample_taps(orig_freq, new_freq)
    c, length = wav.shape
    k = taps.shape[1]
    xp = np.pad(wav.astype(np.float64), ((0, 0), (width, width + o)))
    frames = (xp.shape[1] - k) // o + 1
    win = np.lib.stride_tricks.sliding_window_view(xp, k, axis=1)[:, ::o][:, :frames]     # [C, frames, K]
    out = np.einsum("cfk,jk->cfj", win, taps).reshape(c, frames * n)
    target = -(-n * length // o)
    return out[:, :target].astype(wav.dtype if wav.dtype in (np.float32, np.float64) else np.float32)


def load_wav_array(wav: np.ndarray, n_fft: int = 512, sample_rate: int = 16000) -> np.ndarray:
    """The numeric part of load_wav (data_utils.py:20-29) on an in-memory
    [C, L] array: resample to 16 kHz -> normalize -> STFT -> [F, T, 2C]."""
    if sample_rate != 16000:
        wav = resample_waveform(wav, sample_rate, 16000)
    return to_ref_layout(stft(normalize(wav), n_fft))


# --------------------------------------------------------------------------
# R2: complex <-> magnitude/phase                       transforms.py:111-134
# --------------------------------------------------------------------------
def complex_to_magphase(x: np.ndarray) -> np.ndarray:
    c = x.shape[-1] // 2
    re, im = x[..., :c], x[..., c:]
    mag = np.sqrt(re * re + im * im)
    phase = np.arctan2(im, re)
    return np.concatenate([mag, phase], axis=-1).astype(x.dtype)


def magphase_to_complex(x: np.ndarray) -> np.ndarray:
    c = x.shape[-1] // 2
    mag, phase = x[..., :c], x[..., c:]
    return np.concatenate([mag * np.cos(phase), mag * np.sin(phase)], axis=-1).astype(x.dtype)


# --------------------------------------------------------------------------
# R3: mel filterbank                                      transforms.py:51-77
# --------------------------------------------------------------------------
def _hertz_to_mel(f, dtype):
    """1127 * ln(1 + f/700) in `dtype`.  The logarithm is the correctly rounded
    one (evaluated in float64, rounded once to `dtype`) so that the C++ host code
    of the product and this restatement agree bit for bit; TF's Eigen log may
    differ in the last bit, which moves W entries by up to ~1e-5 (see header)."""
    f = np.asarray(f, dtype=dtype)
    arg = (dtype(1.0) + f / dtype(_MEL_BREAK_FREQUENCY_HERTZ)).astype(dtype)
    ln = np.log(arg.astype(np.float64)).astype(dtype)
    return (dtype(_MEL_HIGH_FREQUENCY_Q) * ln).astype(dtype)


def _linspace(start, stop, num, dtype):
    """start + step*i evaluated in `dtype` (TF LinSpace kernel), last = stop."""
    start, stop = dtype(start), dtype(stop)
    if num == 1:
        return np.array([start], dtype=dtype)
    step = dtype((stop - start) / dtype(num - 1))
    out = (start + step * np.arange(num).astype(dtype)).astype(dtype)
    out[-1] = stop
    return out


def linear_to_mel_weight_matrix(num_mel_bins: int = 20,
                                num_spectrogram_bins: int = 129,
                                sample_rate: float = 8000,
                                lower_edge_hertz: float = 125.0,
                                upper_edge_hertz: float = 3800.0,
                                dtype=np.float32) -> np.ndarray:
    """W[F, M] of tf.signal.linear_to_mel_weight_matrix (PARITY UNPINNED, see
    module header).  All arithmetic in `dtype` (TF default float32)."""
    if num_mel_bins <= 0:
        raise ValueError("num_mel_bins must be positive")
    if lower_edge_hertz < 0.0:
        raise ValueError("lower_edge_hertz must be non-negative")
    if lower_edge_hertz >= upper_edge_hertz:
        raise ValueError("lower_edge_hertz must be < upper_edge_hertz")
    if sample_rate <= 0.0:
        raise ValueError("sample_rate must be positive")
    if upper_edge_hertz > sample_rate / 2:
        raise ValueError("upper_edge_hertz must not exceed the Nyquist frequency")
    dt = np.dtype(dtype).type
    nyquist = dt(sample_rate) / dt(2.0)
    lin = _linspace(0.0, nyquist, num_spectrogram_bins, dt)[1:]  # DC dropped
    bins_mel = _hertz_to_mel(lin, dt)[:, None]  # [F-1, 1]
    edges = _linspace(_hertz_to_mel(lower_edge_hertz, dt),
                      _hertz_to_mel(upper_edge_hertz, dt), num_mel_bins + 2, dt)
    lo, ctr, hi = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
    lower_slopes = (bins_mel - lo) / (ctr - lo)
    upper_slopes = (hi - bins_mel) / (hi - ctr)
    w = np.maximum(dt(0.0), np.minimum(lower_slopes, upper_slopes)).astype(dt)
    return np.concatenate([np.zeros((1, num_mel_bins), dt), w], axis=0)


def magphase_to_mel(num_mel_bins: int = 80, num_spectrogram_bins: int = 257,
                    sample_rate: float = 16000, **kwargs):
    """Closure factory with the reference's signature (transforms.py:51-54)."""
    w = linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins,
                                    sample_rate, **kwargs)

    def _magphase_to_mel(x, y=None):
        x = np.asarray(x)
        x = x[..., : x.shape[-1] // 2]  # drop the phase half (:64)
        if x.ndim not in (3, 4):
            raise ValueError("len(x.shape) must be 3 or 4")
        # tensordot over the freq axis (-3) -> [..., T, C, M]; then M to the front
        out = np.tensordot(x, w.astype(x.dtype), axes=([-3], [0]))
        out = np.moveaxis(out, -1, -3)  # [b, M, T, C] or [M, T, C]
        return out if y is None else (out, y)

    return _magphase_to_mel


# --------------------------------------------------------------------------
# R4/R5: min-max and log                 data_utils.py:37-55, trainer.py:63-77
# --------------------------------------------------------------------------
def safe_div(x, y, eps=EPSILON):
    return x / np.maximum(y, np.asarray(eps, dtype=np.asarray(x).dtype))  # utils.py:114-116


def minmax(x, y=None):
    """Reduce over every axis except 0 (data_utils.py:39): per-sample when
    batched [B,M,T,C], per-mel-row when unbatched [M,T,C] (metrics.py:53)."""
    x = np.asarray(x)
    axis = tuple(range(1, x.ndim))
    x_max = x.max(axis=axis, keepdims=True)
    x_min = x.min(axis=axis, keepdims=True)
    out = safe_div(x - x_min, x_max - x_min).astype(x.dtype)
    return out if y is None else (out, y)


def log_on_mel(mel, labels=None):
    mel = np.asarray(mel)
    out = np.log(mel + mel.dtype.type(EPSILON)).astype(mel.dtype)
    return out if labels is None else (out, labels)


def minmax_log_on_mel(mel, labels=None):
    out = log_on_mel(minmax(mel))
    return out if labels is None else (out, labels)


def log_magphase(specs, labels=None, n_chan=2):
    """log on the first n_chan trailing channels only (transforms.py:80-86)."""
    specs = np.asarray(specs)
    if not np.issubdtype(specs.dtype, np.floating):
        specs = specs.astype(np.float64)
    out = np.concatenate([np.log(specs[..., :n_chan] + specs.dtype.type(EPSILON)),
                          specs[..., n_chan:]], axis=-1)
    return out if labels is None else (out, labels)


def minmax_norm_magphase(specs, labels=None):
    """(x-min)/(max-min+eps) separately for mag and phase halves
    (transforms.py:89-107; note '+eps', not safe_div)."""
    specs = np.asarray(specs)
    c = specs.shape[-1] // 2
    axis = tuple(range(1, specs.ndim))
    parts = []
    for part in (specs[..., :c], specs[..., c:]):
        mx = part.max(axis=axis, keepdims=True)
        mn = part.min(axis=axis, keepdims=True)
        parts.append((part - mn) / (mx - mn + specs.dtype.type(EPSILON)))
    out = np.concatenate(parts, axis=-1)
    return out if labels is None else (out, labels)


# --------------------------------------------------------------------------
# R6: SpecAugment band masks                             transforms.py:12-40
# --------------------------------------------------------------------------
def mask_draw(rng: np.random.Generator, total: int,
              max_mask_size: Optional[int] = None, n_mask: int = 1):
    """Distribution of transforms.py:25-26: size ~ U{0..max_mask_size-1},
    offset ~ U{0..total-size-1}.  Returns int arrays (offsets, sizes)."""
    if max_mask_size is None:
        max_mask_size = total
    offsets, sizes = [], []
    for _ in range(n_mask):
        size = int(rng.integers(0, max_mask_size))
        if total - size <= 0:
            raise ValueError("mask: maxval must be > 0 (max_mask_size > total)")
        offsets.append(int(rng.integers(0, total - size)))
        sizes.append(size)
    return np.asarray(offsets, np.int32), np.asarray(sizes, np.int32)


def mask_apply(specs, axis: int, offsets: Sequence[int], sizes: Sequence[int]):
    """specs * prod_i band_i, band_i zero on [offset_i, offset_i+size_i) along
    `axis`, in specs.dtype (transforms.py:20, :28-40)."""
    specs = np.asarray(specs)
    total = specs.shape[axis]
    m = np.ones(total, dtype=specs.dtype)
    for off, size in zip(offsets, sizes):
        m[off:off + size] = 0
    shape = [1] * specs.ndim
    shape[axis] = total
    return specs * m.reshape(shape)


def augment_apply(specs, t_offsets, t_sizes, f_offsets, f_sizes,
                  time_axis=-2, freq_axis=-3):
    """data_utils.py:58-61 with the random draws made explicit."""
    specs = mask_apply(specs, time_axis, t_offsets, t_sizes)
    return mask_apply(specs, freq_axis, f_offsets, f_sizes)


def random_shift_apply(specs, axis: int, width: int, offset: int):
    """transforms.py:43-47: zero-pad `width` both sides of `axis`, crop the
    original extent starting at `offset` in [0, 2*width]."""
    specs = np.asarray(specs)
    pad = [(0, 0)] * specs.ndim
    pad[axis] = (width, width)
    padded = np.pad(specs, pad)
    sl = [slice(None)] * specs.ndim
    sl[axis] = slice(offset, offset + specs.shape[axis])
    return padded[tuple(sl)]


# --------------------------------------------------------------------------
# R7: channel / filter helpers                         data_utils.py:73-136
# --------------------------------------------------------------------------
def mono_chan(x, y=None):
    if y is not None:
        return x[..., :1] + x[..., 1:], y  # broadcast add, data_utils.py:75
    return x


def stereo_mono(x, y=None):
    out = np.concatenate([x[..., :2], x[..., :1] + x[..., 1:2],
                          x[..., 2:4], x[..., 2:3] + x[..., 3:4]], -1)
    return out if y is None else (out, y)


def stft_filter(filter_num: int):
    def _stft_filter(x, y=None):
        x = np.array(x, copy=True)
        x[1:1 + filter_num] = 0  # bins 1..filter_num of axis 0 (data_utils.py:128-132)
        return x if y is None else (x, y)
    return _stft_filter


def random_merge_aug_apply(x, number: int, factor: np.ndarray):
    """data_utils.py:100-117 with factor[1,1,number-2] ~ U(0.1,0.9) explicit."""
    chan = x.shape[-1] // 2
    if chan != 2:
        raise ValueError("This augment can be used in 2 channel audio")
    real, imag = x[..., :chan], x[..., chan:]
    k = number - chan
    aug_real = factor * np.repeat(real[..., :1], k, -1) \
        + np.sqrt(1 - factor) * np.repeat(real[..., 1:], k, -1)
    real = np.concatenate([real, aug_real], -1)
    imag = np.concatenate([imag, np.repeat(imag[..., :1] + imag[..., 1:], k, -1)], -1)
    return np.concatenate([real, imag], -1)


# --------------------------------------------------------------------------
# R9: label helpers                        data_utils.py:64-70, :85-97, :120-123
# --------------------------------------------------------------------------
def to_frame_labels(x, y):
    return x, np.sum(y, axis=-3)


def avg_pool1d_same(y: np.ndarray, k: int) -> np.ndarray:
    """Keras AveragePooling1D(k, k, padding='same') on [B, T, K]: windows of k
    from 0, the ragged tail window averaged over its valid entries only."""
    b, t, c = y.shape
    n_out = -(-t // k)
    out = np.empty((b, n_out, c), dtype=y.dtype)
    for i in range(n_out):
        out[:, i] = y[:, i * k:(i + 1) * k].mean(axis=1)
    return out


def label_downsample(resolution: int = 32, ref_batch_slice: bool = False):
    """data_utils.py:85-97.  The reference's trailing ``[:resolution]`` slices
    the *batch* axis (harmless for B <= resolution); reproduced only when
    ref_batch_slice=True."""
    def _label_downsample(x, y):
        y_ = avg_pool1d_same(np.asarray(y), resolution)
        y_ = (y_ >= 0.5).astype(y_.dtype)
        if ref_batch_slice:
            y_ = y_[:resolution]
        return x, y_
    return _label_downsample


def preprocess_labels(multiplier):
    """trainer.py:86-94: five passes of tf.nn.avg_pool1d(y, 2, strides=2, 'SAME') * 2 on [B, T, K] (a sum-pool by
    2 whose ragged tail window is averaged over its ONE valid entry and doubled), then * multiplier."""
    def _preprocess(x, y):
        y = np.asarray(y)
        for _ in range(5):
            y = avg_pool1d_same(y, 2) * y.dtype.type(2)
        return x, y * y.dtype.type(multiplier)
    return _preprocess


def to_density_labels(x, y):
    """trainer.py:97-104: [..., V, T, K] -> every voice scaled to unit mass (safe_div, utils.py:114-116), summed
    over the voices axis."""
    y = np.asarray(y)
    y = safe_div(y, np.sum(y, axis=(-2, -1), keepdims=True))
    return x, np.sum(y, axis=-3)


def multiply_label(multiplier):
    """data_utils.py:120-123."""
    def _multiply(x, y):
        return x, y * multiplier
    return _multiply


# --------------------------------------------------------------------------
# remaining transforms.py signatures                    transforms.py:137-195
# --------------------------------------------------------------------------
def phase_vocoder(complex_spec: np.ndarray, rate: float = 1.0) -> np.ndarray:
    """Restatement of transforms.py:137-195 on [F, T, 2C] (re block, im block)."""
    if rate == 1:
        return complex_spec
    spec = np.asarray(complex_spec)
    dt = spec.dtype.type
    freq = spec.shape[0]
    hop_length = freq - 1
    c = spec.shape[-1] // 2

    def angle(s):
        return np.arctan2(s[..., c:], s[..., :c])

    phase_advance = np.linspace(0.0, np.pi * hop_length, freq).astype(spec.dtype).reshape(-1, 1, 1)
    time_steps = np.arange(0, spec.shape[1], rate, dtype=spec.dtype)
    padded = np.pad(spec, [(0, 0), (0, 2), (0, 0)])
    i0 = time_steps.astype(np.int32)
    i1 = (time_steps + 1).astype(np.int32)
    s0, s1 = padded[:, i0], padded[:, i1]
    a0, a1 = angle(s0), angle(s1)
    n0 = np.sqrt(s0[..., :c] ** 2 + s0[..., c:] ** 2)
    n1 = np.sqrt(s1[..., :c] ** 2 + s1[..., c:] ** 2)
    phase_0 = angle(padded[:, :1])
    phase = a1 - a0 - phase_advance
    phase = phase - dt(2 * np.pi) * np.round(phase / dt(2 * np.pi))  # half-to-even, :183
    phase = phase + phase_advance
    phase = np.concatenate([phase_0, phase[:, :-1]], axis=1)
    phase_acc = np.cumsum(phase, axis=1, dtype=spec.dtype)
    alphas = (time_steps % dt(1.0)).reshape(1, -1, 1)
    mag = alphas * n1 + (1 - alphas) * n0
    return np.concatenate([mag * np.cos(phase_acc), mag * np.sin(phase_acc)], axis=-1).astype(spec.dtype)


# --------------------------------------------------------------------------
# R8: sample synthesis, deterministic apply given the draws   pipeline.py:6-110
# --------------------------------------------------------------------------
def merge_complex_specs_apply(background, voices, labels, noises, draws,
                              n_frame=300, n_classes=3, min_ratio=2 / 3,
                              min_noise_ratio=1 / 2, seperate_noise_voice=False):
    """pipeline.py:6-110 (t_axis=1) with every random draw supplied in `draws`:
    {'bg_offset': int, 'n_voices': int, 'v_gain': [..], 'v_offset': [..],
     'n_noises': int, 'n_gain': [..], 'n_offset': [..]}.
    voices: list/array of [F, t_v, 2C]; labels [V, K]; noises list or None.
    Returns (spec [F, n_frame, 2C], label [V, n_frame, K]); with
    seperate_noise_voice (pipeline.py:38-39, :80-81, :104-108) the label is
    (label, only_voice, only_noise): the accepted voices alone (from zeros) and
    the background crop plus the noises."""
    background = np.asarray(background, np.float32)
    bg_frame = background.shape[1]
    reps = (n_frame + bg_frame - 1) // bg_frame
    tiled = np.tile(background, (1, reps, 1))
    o = draws['bg_offset']
    spec = tiled[:, o:o + n_frame].copy()
    only_voice, only_noise = np.zeros_like(spec), spec.copy()
    max_voices = len(voices)
    label = np.zeros((max_voices, n_frame, n_classes), np.float32)
    for v in range(draws['n_voices']):
        voice = np.asarray(voices[v], np.float32)
        v_frame = voice.shape[1]
        l = np.tile(np.asarray(labels[v], np.float32)[None], (v_frame, 1))
        active = (voice.max(axis=(0, 2)) > 0).astype(np.float32)
        l = l * active[:, None]
        pad = n_frame - int(np.float32(min_ratio) * np.float32(v_frame))
        if pad > 0:
            voice = np.pad(voice, [(0, 0), (pad, pad), (0, 0)])
            l = np.pad(l, [(pad, pad), (0, 0)])
        off = draws['v_offset'][v]
        voice = voice[:, off:off + n_frame]
        l = l[off:off + n_frame]
        l3 = np.zeros_like(label)
        l3[v] = l
        no_overlap = np.float32((label + l3).sum(axis=0).max() < 2)
        spec += np.float32(draws['v_gain'][v]) * voice * no_overlap
        if seperate_noise_voice:
            only_voice += np.float32(draws['v_gain'][v]) * voice * no_overlap
        label += l3 * no_overlap
    if noises is not None:
        for n in range(draws['n_noises']):
            noise = np.asarray(noises[n], np.float32)
            ns_frame = noise.shape[1]
            pad = n_frame - int(np.float32(min_noise_ratio) * np.float32(ns_frame))
            if pad > 0:
                noise = np.pad(noise, [(0, 0), (pad, pad), (0, 0)])
            off = draws['n_offset'][n]
            if seperate_noise_voice:
                only_noise += np.float32(draws['n_gain'][n]) * noise[:, off:off + n_frame]
            spec += np.float32(draws['n_gain'][n]) * noise[:, off:off + n_frame]
    if seperate_noise_voice:
        return spec, (label, only_voice, only_noise)
    return spec, label


def wave_frame_active(wav, n_fft, hop):
    """Frame activity of a waveform source [C, L] (waveform-domain twin of pipeline.py:57): frame t is active when
    any sample under the support of its periodic-Hann window, [t*hop - n_fft/2 + 1, t*hop + n_fft/2 - 1] clipped to
    the clip, is non-zero in any channel."""
    wav = np.asarray(wav, np.float32)
    length = wav.shape[1]
    nz = np.concatenate([[0], np.cumsum((wav != 0).any(axis=0))])
    out = np.zeros(1 + length // hop, np.float32)
    for t in range(out.shape[0]):
        lo, hi = max(t * hop - n_fft // 2 + 1, 0), min(t * hop + n_fft // 2 - 1, length - 1)
        out[t] = 1.0 if hi >= lo and nz[hi + 1] - nz[lo] > 0 else 0.0
    return out


def mix_waves_apply(background, voices, labels, noises, draws, n_frame=300, n_classes=3, hop=256, n_fft=1024,
                    min_ratio=2 / 3, min_noise_ratio=1 / 2):
    """Waveform-domain merge_complex_specs (pipeline.py:6-110 with every frame
    quantity multiplied by `hop`): sources are [C, L_i] waveforms, a source of L samples has T = 1 + L // hop frames,
    groups are padded to their longest member as `padded_batch` does.  Returns (wav [C, (n_frame - 1) * hop],
    label [V, n_frame, K]); STFT(wav) equals the spectrum-domain mix of the sources' STFTs on every frame whose
    window crosses no crop / pad / tiling boundary (linearity)."""
    background = np.asarray(background, np.float32)
    out_len = (n_frame - 1) * hop
    lb = background.shape[1]
    idx = (np.int64(draws['bg_offset']) * hop + np.arange(out_len)) % lb
    wav = background[:, idx].copy()
    max_voices = len(voices)
    frames = lambda x: 1 + np.asarray(x).shape[1] // hop  # noqa: E731
    label = np.zeros((max_voices, n_frame, n_classes), np.float32)

    def crop(src, pad, off):
        src = np.asarray(src, np.float32)
        start = (off - pad) * hop
        s = np.arange(out_len) + start
        ok = (s >= 0) & (s < src.shape[1])
        res = np.zeros((src.shape[0], out_len), np.float32)
        res[:, ok] = src[:, s[ok]]
        return res

    v_len = max(frames(v) for v in voices)
    for v in range(draws['n_voices']):
        t_v = frames(voices[v])
        active = wave_frame_active(voices[v], n_fft, hop)
        pad = max(n_frame - int(np.float32(min_ratio) * np.float32(v_len)), 0)
        off = draws['v_offset'][v]
        l = np.zeros((n_frame, n_classes), np.float32)
        for t in range(n_frame):
            fr = off + t - pad
            if 0 <= fr < t_v:
                l[t] = np.asarray(labels[v], np.float32) * active[fr]
        l3 = np.zeros_like(label)
        l3[v] = l
        no_overlap = np.float32((label + l3).sum(axis=0).max() < 2)
        wav += np.float32(draws['v_gain'][v]) * crop(voices[v], pad, off) * no_overlap
        label += l3 * no_overlap
    if noises is not None and len(noises):
        n_len = max(frames(n) for n in noises)
        for n in range(draws['n_noises']):
            pad = max(n_frame - int(np.float32(min_noise_ratio) * np.float32(n_len)), 0)
            wav += np.float32(draws['n_gain'][n]) * crop(noises[n], pad, draws['n_offset'][n])
    return wav, label


# --------------------------------------------------------------------------
# the fused chain the HIP kernel implements
# --------------------------------------------------------------------------
def wav_to_mel(wav: np.ndarray, n_fft: int, hop: int, n_mel: int,
               sample_rate: float = 16000, t_bands=None, f_bands=None,
               dtype=np.float32, **mel_kw) -> np.ndarray:
    """wav[B, C, L] -> mel magnitudes [B, M, T, C] (before min-max/log).

    t_bands / f_bands: optional int arrays [B, n, 2] of (offset, size) zero
    bands along time / linear-frequency, applied to the complex spectrum before
    the magnitude exactly as `augment` does before batching (sj_train.py:108-118).
    """
    wav = np.asarray(wav, dtype=dtype)
    spec = stft(wav, n_fft, hop, dtype=dtype)  # [B, C, F, T]
    mag = np.abs(spec).astype(dtype)
    b = wav.shape[0]
    if t_bands is not None:
        for i in range(b):
            for off, size in np.asarray(t_bands[i]):
                mag[i, :, :, off:off + size] = 0
    if f_bands is not None:
        for i in range(b):
            for off, size in np.asarray(f_bands[i]):
                mag[i, :, off:off + size, :] = 0
    f = n_fft // 2 + 1
    w = linear_to_mel_weight_matrix(n_mel, f, sample_rate, dtype=np.float32, **mel_kw).astype(dtype)
    mel = np.einsum('bcft,fm->bmtc', mag, w, optimize=True)
    return mel.astype(dtype)


# --------------------------------------------------------------------------
# the stated fp32 tolerance of the mel stage, as ONE rule for every shape and input
# --------------------------------------------------------------------------
MEL_REL_TOL = 1e-5        # north_star: mel magnitudes within 1e-5 relative error
MEL_NOISE_ULPS = 4.0      # x eps(fp32) x (rms of the frame's spectrum) x (sum of the band's weights)


def mel_tolerance(wav, n_fft, hop, n_mel, sample_rate=16000, t_bands=None, f_bands=None,
                  rel=MEL_REL_TOL, ulps=MEL_NOISE_ULPS, **mel_kw):
    """(ref, tol): the fp64 mel [B, M, T, C] of wav[B, C, L] and the element-wise bound an fp32 implementation is
    held to,

        |mel - ref| <= rel * |ref|  +  ulps * eps_fp32 * xrms[b, t, c] * sum_k W[k, m]

    First term: north_star's 1e-5 relative error.  Second term: the absolute noise floor of ANY fp32 transform - the
    rounding noise of an n_fft-point fp32 FFT is ~1.2 u x (rms of that frame's spectrum) on every bin, however small
    the bin's own value (scripts/fft_error_model.py: scipy's fp32 pocketfft 1.2-1.7 u, torch.stft 1.1-1.2 u, the HIP
    kernel's operation order 1.1-1.2 u on bins below 2 % of the rms; u = eps / 2), and a mel band passes it on
    weighted by its filter: sum_k W[k, m].  For ordinary values the second term is a few per cent of the first; it
    only matters where a narrow band (the reference's 80 mel over 257 bins has one-bin bands with weights ~0.1) meets
    a bin far below the frame's level.  xrms is the rms over the frame's UNMASKED spectrum (a band zeroed by
    SpecAugment removes signal, not rounding noise); a silent frame gives tol = 0: its mel must be exactly 0.
    (NumPy's own float32 rfft is no yardstick for this floor: numpy 2.x evaluates it in double precision and rounds
    the result once - its complex error equals the rounding of the fp64 result bit for bit.)"""
    wav64 = np.asarray(wav, dtype=np.float64)
    frames = frame_signal(wav64, n_fft, hop) * hann_periodic(n_fft, np.float64)
    mag = np.abs(np.fft.rfft(frames, n=n_fft, axis=-1))  # [B, C, T, F] (frame-major: no transposed copy)
    xrms = np.sqrt(np.mean(mag * mag, axis=-1))  # [B, C, T]
    b = wav64.shape[0]
    if t_bands is not None:
        for i in range(b):
            for off, size in np.asarray(t_bands[i]):
                mag[i, :, off:off + size, :] = 0
                xrms[i, :, off:off + size] = 0  # a masked frame is written as exact zeros
    if f_bands is not None:
        for i in range(b):
            for off, size in np.asarray(f_bands[i]):
                mag[i, :, :, off:off + size] = 0
    w = linear_to_mel_weight_matrix(n_mel, n_fft // 2 + 1, sample_rate, dtype=np.float32, **mel_kw).astype(np.float64)
    ref = np.einsum('bctf,fm->bmtc', mag, w, optimize=True)
    wsum = w.sum(axis=0)  # [M]
    eps = float(np.finfo(np.float32).eps)
    tol = rel * np.abs(ref) + ulps * eps * wsum[None, :, None, None] * np.moveaxis(xrms, 1, -1)[:, None, :, :]
    return ref, tol


def mel_strict_rel_err(mel, ref, tol, rel=MEL_REL_TOL):
    """(worst |mel - ref| / |ref|, fraction of elements covered) over the ORDINARY elements: those where the relative
    term of `mel_tolerance` is at least the noise-floor term (rel |ref| >= tol - rel |ref|, i.e. |ref| >= 0.048 xrms sum W
    with the default constants - ~99 % of the elements of a white-noise input even on the reference's one-bin bands).
    On them north_star's bound is asserted LITERALLY: the result must be <= 1e-5.  The noise-floor term only decides the
    remaining elements - bins far below their frame's level, where the "relative error" of any fp32 transform is
    (absolute rounding noise) / (a value near 0)."""
    ref = np.asarray(ref, np.float64)
    rel_term = rel * np.abs(ref)
    ordinary = (rel_term >= (tol - rel_term)) & (ref != 0)
    if not np.any(ordinary):
        return 0.0, 0.0
    d = np.abs(np.asarray(mel, np.float64) - ref)
    return float((d[ordinary] / np.abs(ref[ordinary])).max()), float(ordinary.mean())


def mel_err_ratio(mel, ref, tol) -> float:
    """max |mel - ref| / tol over the elements (<= 1 passes); elements with tol == 0 must be exact."""
    d = np.abs(np.asarray(mel, np.float64) - ref)
    zero = tol == 0
    if np.any(d[zero] != 0):
        return float('inf')
    return float((d[~zero] / tol[~zero]).max()) if np.any(~zero) else 0.0


def wav_to_logmel(wav, n_fft, hop, n_mel, sample_rate=16000, do_minmax=True,
                  t_bands=None, f_bands=None, dtype=np.float32, **mel_kw):
    mel = wav_to_mel(wav, n_fft, hop, n_mel, sample_rate, t_bands, f_bands, dtype, **mel_kw)
    if do_minmax:
        mel = minmax(mel)
    return log_on_mel(mel)


# --------------------------------------------------------------------------
# eval-path helpers (metrics.py:56-81): tf.signal.frame / overlap_and_add / smoothing
# --------------------------------------------------------------------------
def tf_frame(x: np.ndarray, frame_length: int, frame_step: int, axis: int = -2) -> np.ndarray:
    """tf.signal.frame(..., pad_end=True): ceil(len/step) frames, zero padded at the end."""
    x = np.moveaxis(np.asarray(x), axis, -1)
    n = x.shape[-1]
    num = -(-n // frame_step)
    need = (num - 1) * frame_step + frame_length
    if need > n:
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (need - n,), x.dtype)], -1)
    out = np.stack([x[..., i * frame_step:i * frame_step + frame_length] for i in range(num)], -2)
    return out  # [..., num, frame_length]: the framed axis is moved to the end


def tf_overlap_and_add(frames: np.ndarray, step: int) -> np.ndarray:
    w, length = frames.shape[-2:]
    out = np.zeros(frames.shape[:-2] + ((w - 1) * step + length,), frames.dtype)
    for i in range(w):
        out[..., i * step:i * step + length] += frames[..., i, :]
    return out


def pool1d_same(x: np.ndarray, k: int, mode: str) -> np.ndarray:
    """Keras {Average,Max}Pooling1D(k, strides=1, padding='same') on [T, K]."""
    t = x.shape[0]
    left = (k - 1) // 2
    out = np.empty_like(x)
    for i in range(t):
        lo, hi = max(0, i - left), min(t, i - left + k)
        out[i] = x[lo:hi].max(0) if mode == 'max' else x[lo:hi].mean(0)
    return out


# --------------------------------------------------------------------------
# Device-side draws (challenge_amd/csrc/k_draw.h), restated: Philox4x32-10 + the integer draws of a batch.
# The generator is the published Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
# Random123 known-answer vectors in tests/test_oracle.py); what is drawn follows pipeline.py:29-106 / transforms.py:25-26.
# --------------------------------------------------------------------------
DRAW_SAMPLE, DRAW_VOICE, DRAW_NOISE, DRAW_PERM, DRAW_BAND_T, DRAW_BAND_F = 1, 2, 3, 4, 5, 6
_M32 = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """One Philox4x32-10 block: counter (c0..c3), key (k0, k1) -> four 32-bit words (Python ints)."""
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _M32, p1 & _M32, ((p0 >> 32) ^ c3 ^ k1) & _M32, p0 & _M32
        k0, k1 = (k0 + 0x9E3779B9) & _M32, (k1 + 0xBB67AE85) & _M32
    return c0, c1, c2, c3


def draw_below(word, rng_size):
    return (word * rng_size) >> 32


def draw_unit(word):
    return np.float32(word >> 8) * np.float32(1.0 / 16777216.0)


def stream_perm(i, n, stream, epoch, k0, k1):
    """Element i of epoch `epoch` of stream `stream`: keyed bijection of [0, n) (k_draw.h stream_perm)."""
    if n <= 1:
        return 0
    bits = 1
    while (1 << bits) < n:
        bits += 1
    mask = (1 << bits) - 1
    sh = max(bits // 2, 1)
    h = philox4x32_10(epoch & _M32, (epoch >> 32) & _M32, stream, DRAW_PERM, k0, k1)
    v = i
    while True:
        v = (v + h[0]) & mask
        v = (v * (h[1] | 1)) & mask
        v ^= v >> sh
        v = (v + h[2]) & mask
        v = (v * (h[3] | 1)) & mask
        v ^= v >> sh
        v = (v * 0x9E3779B1) & mask
        v ^= v >> sh
        if v < n:
            return v


def mix_draw_device(bg_T, v_T, n_T, batch, n_frame, max_voices, max_noises, min_ratio, min_noise_ratio, snr, seed, state):
    """The records iris_mix_draw writes for one call, as per-sample dicts in the layout of `merge_draw` (gains as the
    float32 the kernel's formula gives up to the rounding of 10^x).  state = [call counter, bg / voice / noise stream
    positions] (list of 4 ints), advanced in place like the device copy."""
    k0, k1 = seed & _M32, (seed >> 32) & _M32
    ctr, pos_b, pos_v, pos_n = state
    c0, c1 = ctr & _M32, (ctr >> 32) & _M32
    V, Nn = max_voices, (max_noises if n_T is not None and len(n_T) else 0)

    def pick(T, stream, pos):
        n = len(T)
        return stream_perm(pos % n, n, stream, pos // n, k0, k1)

    def padded(frames, ratio):
        pad = n_frame - int(np.float32(ratio) * np.float32(frames))
        return pad, (frames + 2 * pad if pad > 0 else frames)
    out = []
    for b in range(batch):
        rs = philox4x32_10(c0, c1, b * 64, DRAW_SAMPLE, k0, k1)
        bg = pick(bg_T, 0, pos_b + b)
        T = int(bg_T[bg])
        reps = (n_frame + T - 1) // T
        d = {"bg": bg, "bg_offset": draw_below(rs[2], reps * T - n_frame + 1)}
        d["n_voices"] = 1 + draw_below(rs[0], V - 1) if V > 1 else 1
        d["voices"] = [pick(v_T, 1, pos_v + b * V + j) for j in range(V)]
        d["v_len"] = int(max(v_T[i] for i in d["voices"]))
        pad, length = padded(d["v_len"], min_ratio)
        maxval = length - n_frame
        d["v_offset_all"], d["v_gain_all"] = [], []
        for j in range(V):
            rv = philox4x32_10(c0, c1, b * 64 + 1 + j, DRAW_VOICE, k0, k1)
            d["v_offset_all"].append(draw_below(rv[1], maxval) if maxval > 0 else 0)
            d["v_gain_all"].append(np.float32(10.0) ** -(draw_unit(rv[0]) * np.float32(-snr / 10.0)))
        d["v_pad"] = max(pad, 0)
        d["n_noises"], d["noises"], d["n_len"], d["n_offset_all"], d["n_gain_all"], d["n_pad"] = 0, None, 0, [], [], 0
        if Nn:
            d["n_noises"] = draw_below(rs[1], Nn)
            d["noises"] = [pick(n_T, 2, pos_n + b * Nn + j) for j in range(Nn)]
            d["n_len"] = int(max(n_T[i] for i in d["noises"]))
            pad, length = padded(d["n_len"], min_noise_ratio)
            d["n_pad"] = max(pad, 0)
            for j in range(Nn):
                rn = philox4x32_10(c0, c1, b * 64 + 1 + j, DRAW_NOISE, k0, k1)
                d["n_offset_all"].append(draw_below(rn[1], max(length - n_frame, 0) + 1))
                d["n_gain_all"].append(np.float32(10.0) ** -(draw_unit(rn[0]) * np.float32(2.0)))
        out.append(d)
    state[0], state[1], state[2], state[3] = ctr + 1, pos_b + batch, pos_v + batch * V, pos_n + batch * Nn
    return out


def augment_draw_device(batch, n_time, n_t, max_t, n_freq, n_f, max_f, seed, state):
    """(t_bands [B, n_t, 2], f_bands [B, n_f, 2]) of iris_augment_draw; state = [call counter], advanced in place."""
    k0, k1 = seed & _M32, (seed >> 32) & _M32
    c0, c1 = state[0] & _M32, (state[0] >> 32) & _M32
    tb, fb = np.zeros((batch, n_t, 2), np.int32), np.zeros((batch, n_f, 2), np.int32)
    for b in range(batch):
        for arr, n, total, mx, purpose in ((tb, n_t, n_time, max_t, DRAW_BAND_T), (fb, n_f, n_freq, max_f, DRAW_BAND_F)):
            for j in range(n):
                r = philox4x32_10(c0, c1, b * 64 + j, purpose, k0, k1)
                size = draw_below(r[0], mx)
                arr[b, j] = (draw_below(r[1], total - size), size)
    state[0] += 1
    return tb, fb
