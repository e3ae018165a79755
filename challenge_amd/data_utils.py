"""Drop-in counterpart of the reference's data_utils.py on torch tensors
(data_utils.py:9-148).  `load_wav`'s numeric tail (normalize + STFT), `minmax`,
`log_on_mel` and `augment` are the STFT and log ends of the hot path and run as HIP
kernels (ROCm device tensors, no CPU fallback); the channel / label helpers are cheap
device-agnostic torch glue."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import frontend as _fe
from . import transforms as _tr
from .transforms import mask
from .utils import EPSILON, safe_div  # noqa: F401

_SR = 16000


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("load_wav needs a ROCm GPU: the STFT runs as a HIP kernel (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def read_wav_file(wav_fname: str):
    """[chan, samples] float32 in [-1, 1) and the sample rate, from a PCM / float WAV
    (the role of torchaudio.load in data_utils.py:19)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(wav_fname)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    return np.ascontiguousarray(data.T), int(sr)


def stft_array(wav: torch.Tensor, n_fft: int = 512, normalize: bool = False) -> torch.Tensor:
    """Spectrogram(n_fft, power=None) of wav [chan, samples] on a ROCm device, in the
    reference layout [freq, time, chan*2] (re block, im block) (data_utils.py:17-27); `normalize` folds
    data_utils.py:22-23's wav / (10 rms) into the transform (no normalised copy of the waveform)."""
    if wav.dim() != 2:
        raise ValueError("wav must be [chan, samples]")
    plan = _fe.get_plan(wav.device, n_fft, None, 80, _SR, int(wav.shape[0]), 1, int(wav.shape[1]))
    return plan.stft(wav.unsqueeze(0), normalize=normalize)[0]


def load_wav_array(wav, sample_rate: int = _SR, device=None) -> torch.Tensor:
    """The numeric part of load_wav on an in-memory [chan, samples] array."""
    device = _default_device() if device is None else torch.device(device)
    wav = torch.as_tensor(np.asarray(wav, dtype=np.float32) if not isinstance(wav, torch.Tensor) else wav)
    wav = wav.to(device=device, dtype=torch.float32)
    if sample_rate != _SR:   # data_utils.py:20-21: kaldi.resample_waveform(wav, r, 16000), on the device (iris_resample)
        wav = _fe.resample(wav, int(sample_rate), _SR)
    # normalize + STFT as ONE transform launch behind a partial-sums launch (iris_stft with IRIS_F_NORMALIZE): the
    # two-kernel iris_normalize pass over the waveform cost as much as the transform itself.  Equal to
    # stft_array(normalize(wav)) up to the fp32 rounding of the scaled samples (<= 2e-6 of the spectrum's peak)
    return stft_array(wav, 512, normalize=True)


def load_wav(wav_fname: str, device=None) -> torch.Tensor:
    """complex spectrogram [freq, time, chan*2] of a wav file (data_utils.py:9-29)."""
    data, sr = read_wav_file(wav_fname)
    return load_wav_array(data, sr, device)


def resample_waveform(wav: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """torchaudio.compliance.kaldi.resample_waveform as load_wav calls it (data_utils.py:20-21), on a ROCm device tensor."""
    return _fe.resample(wav, orig_freq, new_freq)


def normalize(wav: torch.Tensor) -> torch.Tensor:
    """wav / (10 * rms) with the rms over all channels jointly (data_utils.py:32-34)."""
    if wav.is_cuda:
        return _fe.normalize(wav)
    rms = torch.sqrt(torch.mean(torch.pow(wav, 2))) * 10
    return wav / rms


def minmax(x: torch.Tensor, y=None):
    """(x - min) / max(max - min, EPSILON) over every axis except 0 (data_utils.py:37-47)."""
    x = _fe.minmax_log(x, do_minmax=True, do_log=False)
    if y is not None:
        return x, y
    return x


def log_on_mel(mel: torch.Tensor, labels=None):
    """ln(mel + EPSILON) (data_utils.py:50-55)."""
    mel = _fe.minmax_log(mel, do_minmax=False, do_log=True)
    if labels is not None:
        return mel, labels
    return mel


def minmax_log_on_mel(mel: torch.Tensor, labels=None):
    """minmax + log in one kernel pass (trainer.py:63-77, eval.py:13-27)."""
    mel = _fe.minmax_log(mel, do_minmax=True, do_log=True)
    if labels is not None:
        return mel, labels
    return mel


def augment(specs: torch.Tensor, labels, time_axis: int = -2, freq_axis: int = -3):
    """6 time masks (< 24 frames) then 1 frequency mask (< 16 linear bins) on the complex
    spectrogram (data_utils.py:58-61)."""
    specs = mask(specs, axis=time_axis, max_mask_size=24, n_mask=6)
    specs = mask(specs, axis=freq_axis, max_mask_size=16)
    return specs, labels


def augment_draw(n_time: int, n_freq: int, rng: Optional[np.random.Generator] = None):
    """The random draws of `augment` for one sample: (t_bands [6, 2], f_bands [1, 2])."""
    return _tr.mask_draw(n_time, 24, 6, rng), _tr.mask_draw(n_freq, 16, 1, rng)


def augment_draw_batch(batch: int, n_time: int, n_freq: int, rng: Optional[np.random.Generator] = None):
    """The draws of `augment` for a whole batch, vectorised: (t_bands [B, 6, 2], f_bands [B, 1, 2])."""
    return _tr.mask_draw_batch(batch, n_time, 24, 6, rng), _tr.mask_draw_batch(batch, n_freq, 16, 1, rng)


class DeviceAugmentDraw:
    """The draws of `augment` (6 time masks up to 23 frames, 1 frequency mask up to 15 bins per sample, data_utils.py:58-61)
    made ON THE DEVICE by one small HIP kernel (`iris_augment_draw`: Philox keyed by `seed`, call counter in device
    memory): no host draw, no upload, replayable from a hipGraph.  Optional fixed `stft_filter` band (bins 1..k) appended
    to the frequency bands.  Returns long-lived int32 device tensors (t_bands [B, 6, 2], f_bands [B, 1 (+1), 2])."""

    def __init__(self, device, seed: int = 0, filter_bins: int = 0):
        from . import _native as N
        self._N, self.device = N, torch.device(device)
        N.lib()
        self.seed, self.filter_bins = int(seed) & 0xFFFFFFFFFFFFFFFF, int(filter_bins)
        self.state = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._bufs = {}

    def __call__(self, batch: int, n_time: int, n_freq: int):
        import ctypes as C
        key = (batch, n_time, n_freq)
        if key not in self._bufs:
            nf = 2 if self.filter_bins else 1
            tb = torch.zeros((batch, 6, 2), dtype=torch.int32, device=self.device)
            fb = torch.zeros((batch, nf, 2), dtype=torch.int32, device=self.device)
            if self.filter_bins:
                fb[:, 1, 0], fb[:, 1, 1] = 1, self.filter_bins
            self._bufs[key] = (tb, fb, torch.empty((batch, 1, 2), dtype=torch.int32, device=self.device))
        tb, fb, f1 = self._bufs[key]
        with torch.cuda.device(self.device):
            rc = self._N.lib().iris_augment_draw(batch, n_time, 6, 24, n_freq, 1, 16, self.seed, self.state.data_ptr(),
                                                 tb.data_ptr(), (f1 if self.filter_bins else fb).data_ptr(),
                                                 C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._N.check(rc, "iris_augment_draw")
        if self.filter_bins:
            fb[:, :1].copy_(f1)
        return tb, fb


def to_frame_labels(x, y):
    """[..., n_voices, n_frames, n_classes] -> [..., n_frames, n_classes] (data_utils.py:64-70)."""
    return x, torch.sum(y, dim=-3)


def mono_chan(x, y=None):
    """data_utils.py:73-76, quirks included: with labels it returns x[..., :1] + x[..., 1:]
    (a broadcast add, a true down-mix only for a 2-entry last axis); without labels it is
    the identity."""
    if y is not None:
        return x[..., :1] + x[..., 1:], y
    return x


def stereo_mono(x, y=None):
    """[re0, re1, re0+re1, im0, im1, im0+im1] (data_utils.py:79-82)."""
    out = torch.cat([x[..., :2], x[..., :1] + x[..., 1:2], x[..., 2:4], x[..., 2:3] + x[..., 3:4]], -1)
    if y is None:
        return out
    return out, y


def avg_pool1d_same(y: torch.Tensor, k: int) -> torch.Tensor:
    """Keras AveragePooling1D(k, k, padding='same') on [B, T, K]: the ragged tail window
    is averaged over its valid entries only."""
    yt = y.transpose(1, 2)
    out = torch.nn.functional.avg_pool1d(yt, k, k, ceil_mode=True, count_include_pad=False)
    return out.transpose(1, 2)


def label_downsample(resolution: int = 32, ref_batch_slice: bool = False):
    """Average-pool labels by `resolution`, threshold at 0.5 (data_utils.py:85-97).
    The reference's trailing `[:resolution]` slices the *batch* axis -- harmless while
    batch <= resolution (its default batch is 12) but it would drop rows at BASELINE's
    batch 64, so it is applied only with ref_batch_slice=True (documented deviation)."""
    def _down(y_):
        y_ = avg_pool1d_same(y_, resolution)
        y_ = (y_ >= 0.5).to(y_.dtype)
        return y_[:resolution] if ref_batch_slice else y_

    def _label_downsample(x, y):
        if isinstance(y, (list, tuple)):
            y = (_down(y[0]),) + tuple(y[1:])
        else:
            y = _down(y)
        return x, y
    return _label_downsample


def random_merge_aug_apply(x, number: int, factor):
    """Deterministic half of random_merge_aug (data_utils.py:106-115): `factor` [1, 1, number - 2] is the
    U(0.1, 0.9) draw.  Works on any device (torch slicing / arithmetic on the tensor's own device)."""
    chan = x.shape[-1] // 2
    if chan != 2:
        raise ValueError("This augment can be used in 2 channel audio")
    real, imag = x[..., :chan], x[..., chan:]
    k = number - chan
    factor = torch.as_tensor(factor, dtype=x.dtype, device=x.device)
    aug_real = factor * real[..., :1].repeat_interleave(k, -1) \
        + torch.sqrt(1 - factor) * real[..., 1:].repeat_interleave(k, -1)
    real = torch.cat([real, aug_real], -1)
    imag = torch.cat([imag, (imag[..., :1] + imag[..., 1:]).repeat_interleave(k, -1)], -1)
    return torch.cat([real, imag], -1)


def random_merge_aug(number: int):
    """Extra mixed channels from a 2-channel spectrogram (data_utils.py:100-117): draw + apply."""
    def _random_merge_aug(x, y=None):
        if x.shape[-1] // 2 != 2:
            raise ValueError("This augment can be used in 2 channel audio")
        factor = _tr.get_rng().uniform(0.1, 0.9, size=(1, 1, number - 2))
        out = random_merge_aug_apply(x, number, factor)
        if y is not None:
            return out, y
        return out
    return _random_merge_aug


def multiply_label(multiply_factor):
    def _multiply_label(x, y):
        return x, y * multiply_factor
    return _multiply_label


def stft_filter(filter_num: int):
    """Zero the bins 1..filter_num of axis 0 (data_utils.py:126-136)."""
    def _stft_filter(x, y=None):
        if x.is_cuda:
            x = _fe.mask_apply(x, 0, [[1, filter_num]])
        else:
            x = x.clone()
            x[1:1 + filter_num] = 0
        if y is None:
            return x
        return x, y
    return _stft_filter


def speech_enhancement_preprocess(x, y=None):
    """data_utils.py:139-148 (speech-enhancement model variant; kept for signature parity)."""
    x = x[1:, ..., :x.shape[-1] // 2]
    if y is None:
        return x
    y = (torch.sum(y[0], dim=-3), y[1][1:, ..., :x.shape[-1] // 2], y[2][1:, ..., :x.shape[-1] // 2])
    return x, y
