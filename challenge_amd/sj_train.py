"""Drop-in counterpart of the reference's sj_train.py for the VAD CNN/CRNN path
(sj_train.py:20-255, :402-529): same flags, `make_dataset`, `custom_scheduler`,
`adaptive_clip_grad`, `CustomModel.train_step`, `define_keras_model`, `get_model`,
`main`.  The model is a torch nn.Module on PyTorch-ROCm (MIOpen / hipBLASLt -- not
hand-written, per the north star); the feature frontend feeding it is the HIP library.
Data parallelism (absent in the reference) is one process per GPU with
torch.distributed over RCCL: DistributedDataParallel all-reduces the gradients in
buckets overlapped with backward; AGC and clipvalue then act on the averaged gradients,
identically on every rank.

Out of scope here: model_type 'eff' / 'se' branches
(sj_train.py:258-401), metrics/er_score (metrics.py stays untouched)."""
from __future__ import annotations

import argparse
import csv
import math
import os
import time
from typing import Callable, Optional

import numpy as np
import torch
import torch.nn as nn

from . import data_utils as _du
from . import frontend as _fe
from . import transforms as _tr
from .data_utils import (augment, label_downsample, log_on_mel, minmax, mono_chan, multiply_label,  # noqa: F401
                         random_merge_aug, stereo_mono, stft_filter, to_frame_labels)
from .dataset import AUTOTUNE, Dataset
from .pipeline import make_pipeline
from .transforms import complex_to_magphase, magphase_to_mel
from .utils import label_downsample_model, load_data, sigmoid_focal_crossentropy, unitwise_norm


class ARGS:
    """Same flags and defaults as sj_train.py:20-71, plus the knobs the reference has no
    equivalent for (synthetic data, device, distributed launch)."""

    def __init__(self) -> None:
        self.args = argparse.ArgumentParser()
        a = self.args.add_argument
        a('--name', type=str, default='')
        a('--gpus', type=str, default='-1')
        a('--model', type=int, default=0)
        a('--model_type', type=str, default='vad', choices=['vad', 'eff', 'se'])
        a('--v', type=int, default=1)
        a('--pretrain', type=bool, default=False)
        a('--n_layers', type=int, default=0)
        a('--n_dim', type=int, default=256)
        a('--n_chan', type=int, default=2)
        a('--n_classes', type=int, default=3)
        a('--patience', type=int, default=10)
        # DATA
        a('--mse_multiplier', type=int, default=1)
        a('--datapath', type=str, default='/root/datasets/Interspeech2020/generate_wavs/codes')
        a('--background_sounds', type=str, default='drone_normed_complex_v4.pickle')
        a('--voices', type=str, default='voice_normed_complex_v3.pickle')
        a('--labels', type=str, default='voice_labels_mfc_v3.npy')
        a('--noises', type=str, default='noises_specs_v2.pickle')
        a('--test_background_sounds', type=str, default='test_drone_normed_complex_v2.pickle')
        a('--test_voices', type=str, default='test_voice_normed_complex.pickle')
        a('--test_labels', type=str, default='test_voice_labels_mfc.npy')
        a('--n_mels', type=int, default=80)
        # TRAINING
        a('--optimizer', type=str, default='adam', choices=['adam', 'sgd', 'rmsprop', 'adabelief'])
        a('--lr', type=float, default=1e-3)
        a('--end_lr', type=float, default=1e-4)
        a('--lr_power', type=float, default=0.5)
        a('--lr_div', type=float, default=2)
        a('--clipvalue', type=float, default=0.01)
        a('--epochs', type=int, default=300)
        a('--batch_size', type=int, default=12)
        a('--n_frame', type=int, default=512)
        a('--steps_per_epoch', type=int, default=100)
        a('--l1', type=float, default=0)
        a('--l2', type=float, default=1e-6)
        a('--loss', type=str, default='BCE')
        # AUGMENTATION
        a('--snr', type=float, default=-20)
        a('--max_voices', type=int, default=7)
        a('--max_noises', type=int, default=2)
        # not in the reference
        a('--synthetic', action='store_true', help='synthetic sources instead of the pickled datasets')
        a('--online_stft', action='store_true',
          help='keep (synthetic) waveform corpora on the device, mix them before the STFT (WaveMixer) and run the '
               'fused HIP frontend on line instead of mixing pre-computed spectra (make_wave_dataset)')
        a('--per_sample_pipeline', action='store_true',
          help='build samples one at a time with the tf.data-shaped graph (make_dataset) instead of the '
               'batched on-device synthesis (make_device_dataset), which is the default on a GPU')
        a('--host_draws', action='store_true',
          help='draw the random half of every batch (source choice, offsets, gains, SpecAugment bands) on the host and upload '
               'it, instead of on the device (iris_mix_draw / iris_augment_draw, the default on a GPU: no upload, so the '
               'host is never held behind the previous step)')
        a('--no_clipvalue_after_agc', action='store_true',
          help="skip Adam's element-wise clipvalue (TF < 2.4 behaviour of the custom train_step)")
        a('--validation_steps', type=int, default=16)

    def get(self, argv=None):
        return self.args.parse_args(argv)


# ---------------------------------------------------------------------------
# dataset assembly                                            sj_train.py:74-130
# ---------------------------------------------------------------------------
def complex_to_mel(n_mels: int, num_spectrogram_bins: int = 257, sample_rate: float = 16000, **kwargs):
    """complex_to_magphase + magphase_to_mel as ONE kernel (sj_train.py:119-120 maps them
    one after the other; the phase computed by the first is discarded by the second)."""
    to_mel = magphase_to_mel(n_mels, num_spectrogram_bins, sample_rate, **kwargs)
    n_fft = 2 * (num_spectrogram_bins - 1)
    plans = {}

    def _complex_to_mel(x, y=None, t_bands=None, f_bands=None):
        """t_bands / f_bands ([B, n, 2] (offset, size), optional): SpecAugment / stft_filter bands
        zeroed in the complex spectrum first (== `augment` / `stft_filter` mapped before this stage)."""
        if not x.is_cuda or n_fft not in (256, 512, 1024, 2048):
            if t_bands is not None:
                x = _tr.mask_apply(x, -2, t_bands)
            if f_bands is not None:
                x = _tr.mask_apply(x, -3, f_bands)
            out = to_mel(complex_to_magphase(x))
        else:
            chan = x.shape[-1] // 2
            key = (x.device.index, chan)
            plan = plans.get(key)
            if plan is None or plan.max_batch < x.shape[0]:
                plan = _fe.FrontendPlan(n_fft, None, n_mels, sample_rate, chan, max(int(x.shape[0]), 1), n_fft,
                                        x.device, mel_matrix=to_mel.mel_matrix)
                plans[key] = plan
            out = plan.magmel(x.float(), is_magphase=False, t_bands=t_bands, f_bands=f_bands)
        return out if y is None else (out, y)
    return _complex_to_mel


def synthetic_sources(n_chan: int = 2, n_classes: int = 3, freq: int = 257, n_bg: int = 8, n_voice: int = 24,
                      n_noise: int = 12, seed: int = 0):
    """Random stand-ins for the pickled datasets of sj_train.py:79-89: lists of
    [freq, t_i, 2*chan] float32 spectra and integer class labels."""
    rng = np.random.default_rng(seed)
    backgrounds = [rng.standard_normal((freq, int(rng.integers(300, 900)), 2 * n_chan)).astype(np.float32) * 0.1
                   for _ in range(n_bg)]
    voices = []
    for _ in range(n_voice):
        t = int(rng.integers(40, 200))
        v = np.abs(rng.standard_normal((freq, t, 2 * n_chan))).astype(np.float32)
        v[:, int(t * 0.8):] = 0  # trailing silence, as padded voices have
        voices.append(v)
    labels = rng.integers(0, n_classes, size=n_voice)
    noises = [rng.standard_normal((freq, int(rng.integers(20, 120)), 2 * n_chan)).astype(np.float32) * 0.3
              for _ in range(n_noise)]
    return backgrounds, voices, labels, noises


def _load_sources(config, training, n_classes, sources):
    """The pickled corpora of sj_train.py:79-89 (or `sources` / --synthetic stand-ins) with one-hot labels."""
    if sources is None and getattr(config, 'synthetic', False):
        sources = synthetic_sources(2, n_classes, seed=0 if training else 1)
    if sources is None:
        if not os.path.exists(config.datapath):
            config.datapath = ''
        if training:
            backgrounds = load_data(os.path.join(config.datapath, config.background_sounds))
            voices = load_data(os.path.join(config.datapath, config.voices))
            labels = load_data(os.path.join(config.datapath, config.labels))
        else:
            backgrounds = load_data(os.path.join(config.datapath, config.test_background_sounds))
            voices = load_data(os.path.join(config.datapath, config.test_voices))
            labels = load_data(os.path.join(config.datapath, config.test_labels))
        if labels.max() - 1 != config.n_classes:
            labels //= 10
        noises = load_data(os.path.join(config.datapath, config.noises))
    else:
        backgrounds, voices, labels, noises = sources
    labels = np.eye(n_classes, dtype='float32')[np.asarray(labels)]  # to one-hot vectors
    return backgrounds, voices, labels, noises


def _label_tail(pipeline, config):
    """The stages after the mel features (sj_train.py:121-129)."""
    if 'nominmax' not in config.name:
        pipeline = pipeline.map(_du.minmax_log_on_mel)  # :121-123 fused
    else:
        pipeline = pipeline.map(log_on_mel)
    if config.v in label_downsample_model:
        pipeline = pipeline.map(label_downsample(32))
    elif config.v == 5:
        pipeline = pipeline.map(label_downsample(config.n_frame // (config.n_frame * 256 // 16000)))
    if config.loss.upper() in ('MSE', 'MAE'):
        pipeline = pipeline.map(multiply_label(config.mse_multiplier))
    return pipeline.prefetch(AUTOTUNE)


def make_dataset(config, training=True, n_classes=3, sources=None):
    """Stage order of sj_train.py:74-130.  `sources` = (backgrounds, voices, labels, noises)
    overrides the pickle files (used with --synthetic and by the tests)."""
    backgrounds, voices, labels, noises = _load_sources(config, training, n_classes, sources)

    pipeline = make_pipeline(backgrounds, voices, labels, noises, n_frame=config.n_frame,
                             max_voices=config.max_voices, max_noises=config.max_noises, n_classes=n_classes,
                             snr=config.snr, min_ratio=1,
                             seperate_noise_voice=config.model_type == 'se' and config.v == 9)
    if config.model_type == 'se' and config.v == 9:
        raise NotImplementedError("model_type 'se' is outside the accelerated path (SURVEY.md section 2)")
    pipeline = pipeline.map(to_frame_labels)
    if training:
        pipeline = pipeline.map(augment)
    if config.n_chan == 1:
        pipeline = pipeline.map(mono_chan)
    elif config.n_chan == 3:
        pipeline = pipeline.map(stereo_mono)
    elif config.n_chan > 3:
        pipeline = pipeline.map(random_merge_aug(config.n_chan))
    if 'filter' in config.name:
        pipeline = pipeline.map(stft_filter(int(round(200 / (16000 / 256)))))
    pipeline = pipeline.batch(config.batch_size, drop_remainder=False)
    n_bins = int(np.asarray(backgrounds[0]).shape[0])
    pipeline = pipeline.map(complex_to_mel(config.n_mels, n_bins))  # :119-120 fused
    return _label_tail(pipeline, config)


def make_device_dataset(config, training=True, n_classes=3, sources=None, device=None, seed=None, device_draw=False):
    """MI355X-native `make_dataset`: same stages, same outputs (sj_train.py:74-130), but a whole
    batch at a time on the device.  The corpora stay resident in HBM; `DeviceMixer` synthesises the
    batch in two launches (merge_complex_specs, pipeline.py:6-110); SpecAugment and `stft_filter`
    reach the mel kernel as band descriptors instead of being multiplied into the 135 MB complex
    batch (they zero time / frequency ranges, which commutes with the per-bin channel mixes and with
    the magnitude).  Yields (x [B, n_mels, n_frame, C], y) forever, like the repeated reference graph.
    device_draw=True: the random half of a batch (sources, offsets, gains, SpecAugment bands) is drawn by two small HIP
    kernels on the device (`iris_mix_draw`, `iris_augment_draw`) instead of NumPy on the host - no table upload, the
    host only enqueues launches."""
    from .mixer import DeviceMixer
    backgrounds, voices, labels, noises = _load_sources(config, training, n_classes, sources)
    if config.model_type == 'se' and config.v == 9:
        raise NotImplementedError("model_type 'se' is outside the accelerated path (SURVEY.md section 2)")
    mixer = DeviceMixer(backgrounds, voices, labels, noises, n_frame=config.n_frame, max_voices=config.max_voices,
                        max_noises=config.max_noises, n_classes=n_classes, device=device, snr=config.snr,
                        min_ratio=1, seed=seed)
    rng = np.random.default_rng(None if seed is None else seed + 1)
    filter_bins = int(round(200 / (16000 / 256))) if 'filter' in config.name else 0
    band_draw = None
    if device_draw:
        mixer.enable_device_draw(0 if seed is None else seed)
        band_draw = _du.DeviceAugmentDraw(mixer.device, (0 if seed is None else seed) + 1, filter_bins) if training else None
    to_mel = complex_to_mel(config.n_mels, mixer.n_bins)
    chan_map = None
    if config.n_chan == 1:
        chan_map = mono_chan
    elif config.n_chan == 3:
        chan_map = stereo_mono
    elif config.n_chan > 3:
        chan_map = random_merge_aug(config.n_chan)

    def gen():
        while True:
            x, y = to_frame_labels(*mixer.mix(config.batch_size))
            b = int(x.shape[0])
            tb = fb = None
            if band_draw is not None:
                tb, fb = band_draw(b, config.n_frame, mixer.n_bins)  # filter band included
            elif training:  # `augment` (data_utils.py:58-61): 6 time masks, 1 frequency mask per sample
                tb, fb = _du.augment_draw_batch(b, config.n_frame, mixer.n_bins, rng)
            if filter_bins and band_draw is None:  # stft_filter (data_utils.py:126-136): bins 1..k
                flt = np.tile(np.array([[[1, filter_bins]]], np.int32), (b, 1, 1))
                fb = flt if fb is None else np.concatenate([fb, flt], axis=1)
            if chan_map is not None:
                x, y = chan_map(x, y)
            yield to_mel(x, y, t_bands=tb, f_bands=fb)

    return _label_tail(Dataset.from_generator(gen), config)


def synthetic_wave_sources(n_chan: int = 2, n_classes: int = 3, hop: int = 256, n_bg: int = 8, n_voice: int = 24,
                           n_noise: int = 12, seed: int = 0):
    """Waveform stand-ins of the same shape statistics as `synthetic_sources` (frame counts x hop samples):
    lists of [chan, L_i] float32 waveforms and integer class labels."""
    rng = np.random.default_rng(seed)
    backgrounds = [rng.standard_normal((n_chan, hop * int(rng.integers(300, 900)))).astype(np.float32) * 0.1
                   for _ in range(n_bg)]
    voices = []
    for _ in range(n_voice):
        t = int(rng.integers(40, 200))
        v = rng.standard_normal((n_chan, hop * t)).astype(np.float32) * 0.3
        v[:, hop * int(t * 0.8):] = 0  # trailing silence, as padded voices have
        voices.append(v)
    labels = rng.integers(0, n_classes, size=n_voice)
    noises = [rng.standard_normal((n_chan, hop * int(rng.integers(20, 120)))).astype(np.float32) * 0.3
              for _ in range(n_noise)]
    return backgrounds, voices, labels, noises


def make_wave_dataset(config, training=True, n_classes=3, sources=None, device=None, seed=None, n_fft=512, hop=256,
                      sample_rate=16000, device_draw=False):
    """`make_device_dataset` from WAVEFORMS (waveform-domain variant): the corpora
    stay resident in HBM as [chan, L_i] waveforms, `WaveMixer` mixes a batch before the STFT (which is linear), and
    the fused kernel takes it from there - STFT, SpecAugment / `stft_filter` bands, mel, min-max, log in one pass;
    no spectrum is ever materialised.  Same stages as sj_train.py:74-130 otherwise; `sources` =
    (background waveforms, voice waveforms, labels, noise waveforms).  n_fft defaults to the reference's 512 (F = 257).
    INTENTIONAL DIVERGENCE at n_chan == 1 with stereo corpora: here the two channels are summed BEFORE the STFT - a true
    down-mix, features [B, M, T, 1].  The reference's `mono_chan` (data_utils.py:73-76), which `make_device_dataset`
    reproduces quirk and all, is the broadcast `x[..., :1] + x[..., 1:]` on the 4-entry re / im axis: 3 entries, i.e.
    2 magnitude channels [B, M, T, 2].  So `--online_stft` and the default path feed the model different features
    and channel counts for stereo corpora at n_chan == 1; at n_chan == 2 (the reference's default) they agree up to
    the boundary frames of the waveform-domain mix.  The augmenting maps of n_chan > 2 mix spectra with per-bin
    factors and are not available here.  device_draw=True: sources / offsets / gains / SpecAugment bands are drawn on
    the device (`iris_mix_draw`, `iris_augment_draw`), as in `make_device_dataset`."""
    from .mixer import WaveMixer
    if sources is None:
        sources = synthetic_wave_sources(2, n_classes, hop, seed=0 if training else 1)
    backgrounds, voices, labels, noises = sources
    labels = np.eye(n_classes, dtype='float32')[np.asarray(labels)]
    if config.model_type == 'se' and config.v == 9:
        raise NotImplementedError("model_type 'se' is outside the accelerated path (SURVEY.md section 2)")
    if config.n_chan not in (1, 2):
        raise NotImplementedError("make_wave_dataset: n_chan 1 (mono sum) or 2; the augmenting channel maps mix spectra")
    mixer = WaveMixer(backgrounds, voices, labels, noises, n_frame=config.n_frame, n_fft=n_fft, hop=hop,
                      max_voices=config.max_voices, max_noises=config.max_noises, n_classes=n_classes, device=device,
                      snr=config.snr, min_ratio=1, seed=seed)
    rng = np.random.default_rng(None if seed is None else seed + 1)
    length = (config.n_frame - 1) * hop
    plan = _fe.FrontendPlan(n_fft, hop, config.n_mels, sample_rate, config.n_chan, config.batch_size, length, mixer.device)
    filter_bins = int(round(200 / (16000 / 256))) if 'filter' in config.name else 0
    do_minmax = 'nominmax' not in config.name
    band_draw = None
    if device_draw:
        mixer.enable_device_draw(0 if seed is None else seed)
        band_draw = _du.DeviceAugmentDraw(mixer.device, (0 if seed is None else seed) + 1, filter_bins) if training else None

    def gen():
        while True:
            wav, y = mixer.mix(config.batch_size)
            _, y = to_frame_labels(None, y)
            if config.n_chan == 1 and wav.shape[1] == 2:
                wav = wav[:, :1] + wav[:, 1:]            # true down-mix (NOT the reference's broadcast mono_chan: see docstring)
            b = int(wav.shape[0])
            tb = fb = None
            if band_draw is not None:
                tb, fb = band_draw(b, config.n_frame, plan.n_bins)
            elif training:
                tb, fb = _du.augment_draw_batch(b, config.n_frame, plan.n_bins, rng)
            if filter_bins and band_draw is None:
                flt = np.tile(np.array([[[1, filter_bins]]], np.int32), (b, 1, 1))
                fb = flt if fb is None else np.concatenate([fb, flt], axis=1)
            yield plan.wav_to_logmel(wav.contiguous(), t_bands=tb, f_bands=fb, minmax=do_minmax, log=True), y

    pipeline = Dataset.from_generator(gen)
    if config.v in label_downsample_model:
        pipeline = pipeline.map(label_downsample(32))
    elif config.v == 5:
        pipeline = pipeline.map(label_downsample(config.n_frame // (config.n_frame * 256 // 16000)))
    if config.loss.upper() in ('MSE', 'MAE'):
        pipeline = pipeline.map(multiply_label(config.mse_multiplier))
    return pipeline.prefetch(AUTOTUNE)


class WaveFrontend:
    """MI355X-native on-line variant of the chain: waveforms [B, C, L] on the device ->
    SpecAugment bands drawn per sample -> fused HIP kernel (STFT, magnitude, masks, mel,
    min-max, log) -> [B, M, T, C].  Equivalent to load_wav + augment + complex_to_magphase
    + magphase_to_mel + minmax + log_on_mel without materialising the spectrum."""

    def __init__(self, n_fft=1024, hop=256, n_mels=64, sample_rate=16000, n_chan=1, batch=64, length=130816,
                 device=None, training=True, filter_bins: int = 0, do_minmax: bool = True,
                 device_draw: bool = False, seed: int = 0):
        self.plan = _fe.FrontendPlan(n_fft, hop, n_mels, sample_rate, n_chan, batch, length, device)
        self.training, self.filter_bins, self.do_minmax = training, filter_bins, do_minmax
        self.device_draw = device_draw
        self.rng = np.random.default_rng(seed)
        self._gen = torch.Generator(device=self.plan.device).manual_seed(seed)

    def draw_bands(self, batch: int, n_time: int):
        """Host draw (NumPy Generator): exact integer distributions of transforms.py:25-26."""
        return _du.augment_draw_batch(batch, n_time, self.plan.n_bins, self.rng)

    def draw_bands_device(self, batch: int, n_time: int):
        """Device draw, no host round trip: size ~ U{0..max-1}, offset = floor(U[0,1) * (total - size))
        (same support as the reference's integer draw; probabilities equal up to fp32 rounding)."""
        dev = self.plan.device

        def draw(total, max_size, n):
            size = torch.randint(0, max_size, (batch, n), device=dev, generator=self._gen)
            off = (torch.rand((batch, n), device=dev, generator=self._gen) * (total - size)).floor().to(torch.int64)
            off = torch.minimum(off, total - size - 1)
            return torch.stack([off, size], dim=-1).to(torch.int32)
        return draw(n_time, 24, 6), draw(self.plan.n_bins, 16, 1)

    def __call__(self, wav: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, n_time = wav.shape[0], self.plan.num_frames(wav.shape[2])
        tb = fb = None
        if self.training:
            tb, fb = self.draw_bands_device(b, n_time) if self.device_draw else self.draw_bands(b, n_time)
        if self.filter_bins:
            if isinstance(fb, torch.Tensor) or (fb is None and self.device_draw):
                flt = torch.tensor([[[1, self.filter_bins]]], dtype=torch.int32, device=self.plan.device).expand(b, 1, 2)
                fb = flt if fb is None else torch.cat([fb, flt], dim=1)
            else:
                flt = np.tile(np.array([[[1, self.filter_bins]]], np.int32), (b, 1, 1))
                fb = flt if fb is None else np.concatenate([fb, flt], axis=1)
        return self.plan.wav_to_logmel(wav, minmax=self.do_minmax, log=True, t_bands=tb, f_bands=fb, out=out)


# ---------------------------------------------------------------------------
# schedule, AGC                                             sj_train.py:133-155
# ---------------------------------------------------------------------------
def custom_scheduler(d_model, warmup_steps=4000, lr_div=2):
    """lr(step) = d_model^-0.5 * min((step+1)^-0.5, (step+1) * warmup^-1.5) / lr_div, with
    `step` the EPOCH index as Keras' LearningRateScheduler passes it (sj_train.py:501-503)."""
    d_model = float(d_model)

    def _scheduler(step):
        step = float(step + 1)
        arg1 = step ** -0.5
        arg2 = step * (warmup_steps ** -1.5)
        return d_model ** -0.5 * min(arg1, arg2) / lr_div
    return _scheduler


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
        table = np.empty((rep.shape[0], 3), np.int64)
        if fast:
            table[:, 0] = np.array([p.data_ptr() for p in fast], np.int64)[rep] + offs
            table[:, 1] = np.array([p.grad.data_ptr() for p in fast], np.int64)[rep] + offs
            table[:, 2] = length
        recs = [table]
        table = np.concatenate(recs) if recs else np.zeros((0, 3), np.int64)
        # pinned staging + asynchronous copy: legal while a hipGraph is being captured (it becomes a copy node of the graph).
        # Under capture the buffers must already exist (`reserve`, called by GraphedTrainStep before the capture starts:
        # allocating pinned memory inside a capture invalidates it); the staging buffer stays alive for as long as a captured
        # copy may replay from it.
        reserved = getattr(self, '_reserved', None)
        if reserved is not None and reserved[0].shape[0] >= table.shape[0]:
            # `reserve` sized the buffers for EVERY parameter having a gradient in a layout the kernel takes; a parameter
            # without a gradient or on the torch path (`_slow`) only makes the table shorter: fill a prefix and hand the
            # kernel the actual row count (no allocation inside a capture whatever the row count turns out to be)
            n = int(table.shape[0])
            host, dev_table = reserved[0][:n], reserved[1][:n]
            self._reserved = None
            host.numpy()[...] = table
            dev_table.copy_(host, non_blocking=True)
            self._host_table, self._table = host, dev_table
        else:
            host = torch.from_numpy(table)
            if self.params[0].is_cuda:
                host = host.pin_memory()
                self._host_table = host
            self._table = host.to(self.params[0].device, non_blocking=True)
        self._sig = self._signature()

    def _signature(self):
        """(parameter address, gradient address, gradient strides) per parameter: a gradient buffer handed back at the same
        address in another layout must not reuse a table built for the old one (fast / slow classification, row stride)."""
        return tuple((p.data_ptr(),) + ((-1, ()) if p.grad is None else (p.grad.data_ptr(), tuple(p.grad.stride())))
                     for p in self.params)

    def reserve(self) -> None:
        """Allocate the table and its pinned staging buffer NOW (outside any capture), sized for every parameter having a
        gradient in a layout the kernel takes; the next `_build` fills them in place."""
        rows = sum(self._rows_of(p)[0] for p in self.params)
        host = torch.empty((rows, 3), dtype=torch.int64)
        if self.params[0].is_cuda:
            host = host.pin_memory()
        self._reserved = (host, torch.empty((rows, 3), dtype=torch.int64, device=self.params[0].device))

    def freeze(self) -> None:
        """After a hipGraph capture: the table and its pinned staging buffer are referenced by the graph and must never be
        rebuilt; any later call of this object raises instead."""
        self._frozen = True

    def __call__(self, clip_factor=0.01, eps=1e-3, clipvalue=None):
        import ctypes as C
        if getattr(self, '_frozen', False):
            raise RuntimeError("FusedAGC: this instance belongs to a captured hipGraph (GraphedTrainStep) and cannot be "
                               "called eagerly; eager steps use the model's own instance")
        sig = self._signature()
        if sig != self._sig:
            hit = self._cache.get(sig)
            if hit is not None:
                self._sig, self._table, self._slow, self._host_table = sig, hit[0], hit[1], hit[2]
            else:
                self._build()
                if len(self._cache) >= 8:
                    self._cache.clear()
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


# ---------------------------------------------------------------------------
# model                                                     sj_train.py:191-255
# ---------------------------------------------------------------------------
# Training-mode Conv2D bias + BatchNorm + ReLU through the HIP kernels iris_bn_* (two passes over the activation each way
# instead of seven forward / nine backward); IRIS_FUSED_BN=0 keeps the stock torch / MIOpen ops.
FUSED_BN_RELU = os.environ.get("IRIS_FUSED_BN", "1") != "0"
FUSED_FC_BN = os.environ.get("IRIS_FUSED_FC_BN", "1") != "0"      # Dense + BatchNorm1d + ReLU through the same passes
FUSED_BN_POOL = os.environ.get("IRIS_FUSED_BN_POOL", "1") != "0"  # a block's MaxPool inside its last layer's passes


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
    a bias gradient that autograd adopts from a slice is zero and stays zero."""

    def __init__(self):
        self._state = {}   # (device index, dtype) -> [buffer, offset, wanted]
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


# BatchNorm's num_batches_tracked counters of the layers whose fused passes ran, bumped by ONE _foreach_add_ at the end of
# CustomModel.forward instead of one launch per layer (None outside that forward: the layers then bump their own)
_NBT_PENDING = None


def _count_batch(bn) -> None:
    if _NBT_PENDING is not None:
        _NBT_PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


_ZERO_POOL = _ZeroPool()
ZERO_POOL = os.environ.get("IRIS_ZERO_POOL", "1") != "0"


def _zeros(n: int, dtype: torch.dtype, device: torch.device, kind: str = "scratch") -> torch.Tensor:
    return _ZERO_POOL.take(int(n), dtype, device, kind) if ZERO_POOL else torch.zeros(int(n), dtype=dtype, device=device)


class _FusedBiasBNReLU(torch.autograd.Function):
    """y = relu(batch_norm(z + conv_bias)) in training mode on a channels_last fp32 convolution output z (sj_train.py:191-201),
    with `pool` also the block's MaxPool2d(2, 2, ceil_mode=True) behind it (the full-size y and dy then never exist).
    The bias never touches the activation: batch normalisation subtracts the batch mean, so y does not depend on it (it
    only shifts the running mean, which iris_bn_relu_apply accounts for) and its gradient is identically zero."""

    @staticmethod
    def forward(ctx, z, conv_bias, gamma, beta, running_mean, running_var, eps, momentum, pool=False):
        import ctypes as C
        from . import _native as N
        b, c, h, w = (int(v) for v in z.shape)
        rows = b * h * w
        dev = z.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        sums = _zeros(N.lib().iris_bn_sums_len(c), torch.float64, dev)
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
            N.check(lib.iris_bn_stats(z.data_ptr(), rows, c, sums.data_ptr(), stream), "iris_bn_stats")
            if pool:
                N.check(lib.iris_bn_relu_pool_apply(z.data_ptr(), y.data_ptr(), b, h, w, c, *tail), "iris_bn_relu_pool_apply")
            else:
                N.check(lib.iris_bn_relu_apply(z.data_ptr(), y.data_ptr(), rows, c, *tail), "iris_bn_relu_apply")
        ctx.save_for_backward(z, gamma, beta, save_mean, save_rstd)  # y is not needed: the mask is recomputed from z
        ctx.has_bias = conv_bias is not None
        ctx.pool = bool(pool)
        ctx.mark_non_differentiable(running_mean, running_var)
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
        return dz, dbias, dgamma, dbeta, None, None, None, None, None


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


FUSED_CONV0 = os.environ.get("IRIS_FUSED_CONV0", "1") != "0"


def _is_first_layer_conv(conv, x) -> bool:
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    co = conv.out_channels
    return (conv.in_channels in (1, 2) and pair(conv.kernel_size) == (3, 3) and pair(conv.padding) == (1, 1)
            and pair(conv.stride) == (1, 1) and pair(conv.dilation) == (1, 1) and conv.groups == 1
            and co % 4 == 0 and co <= 256 and 1024 % co == 0 and not x.requires_grad and x.dim() == 4
            and x.shape[3] <= 2048)


# Training: the bare 3x3 convolutions of blocks 2-5 - forward and backward-data - as Winograd F(2x2, 3x3) on the fp32 matrix
# cores (iris_conv3x3_wino) instead of MIOpen's implicit GEMMs, reading and writing channels_last where the weight-gradient
# kernel and the BatchNorm passes read it (that layout costs the kernel 14 % against its own chunked one).  Wherever the kernel's shape rule
# holds (8 | input channels, 64 | output channels - per direction, the backward-data pass swaps them) and the wider side has
# >= 64 channels: every layer of blocks 2-5 forward, all but block 2's first backward.  Thresholds per direction by environment
# (the measured step is flat within noise between 64 and 128: profiles/r5/c4_wino_train_ab.log); IRIS_WINO_TRAIN=0 keeps MIOpen
# everywhere.  13.4-13.7 -> 12.2-12.8 ms per batch-64 step.
WINO_TRAIN = os.environ.get("IRIS_WINO_TRAIN", "1") != "0"
WINO_TRAIN_MIN_C_FWD = int(os.environ.get("IRIS_WINO_TRAIN_MIN_C_FWD", "64"))
WINO_TRAIN_MIN_C_BWD = int(os.environ.get("IRIS_WINO_TRAIN_MIN_C_BWD", "64"))
# the weight gradient of the same layers as Winograd on the fp32 MFMA too (k_conv_wino_wrw.h; channel counts multiples of 32):
# 1.6-2.0x MIOpen's weight-gradient kernels on the step's shapes, deterministic; IRIS_WINO_TRAIN_WRW=0 keeps MIOpen's
WINO_TRAIN_WRW = WINO_TRAIN and os.environ.get("IRIS_WINO_TRAIN_WRW", "1") != "0"
# block 1's 32 -> 32 layer: forward and backward-data by the inference engine's implicit-GEMM kernel (k_conv_c32.h) without
# bias / ReLU instead of CK's / MIOpen's kernels (431 + ~470 us per step); IRIS_C32_TRAIN=0 keeps those
C32_TRAIN = WINO_TRAIN and os.environ.get("IRIS_C32_TRAIN", "1") != "0"


class _WinoConv3x3(torch.autograd.Function):
    """z = conv2d(x, weight, padding=1) for channels_last fp32 tensors.  forward (`fwd`): iris_conv3x3_wino on the weights packed
    on the device this step, else MIOpen; backward: dx (`bwd`) by the same kernel on the transposed / flipped weights, else
    MIOpen; dW (`wrw`) by iris_conv3x3_wino_wrw, else MIOpen's weight-gradient kernel (aten.convolution_backward).
    `fwd` / `bwd` == 'c32': the 32 -> 32 layer of block 1 - forward and backward-data by the implicit-GEMM kernel of the
    inference engine without its bias / ReLU (iris_conv3x3_c32; the backward pass reads the weight transposed and flipped)."""

    @staticmethod
    def forward(ctx, x, weight, fwd=True, bwd=True, wrw=False):
        if fwd == 'c32':
            z = _fe.conv3x3_c32(x, weight)
        elif fwd:
            z = _fe.conv3x3_wino(x, _fe.wino_pack_weights_device(weight), None, int(weight.shape[0]), out_nhwc=True, relu=False)
        else:
            z = torch.nn.functional.conv2d(x, weight, None, 1, 1)
        ctx.save_for_backward(x, weight)
        ctx.wino_bwd = bwd if bwd == 'c32' else bool(bwd)
        ctx.wino_wrw = bool(wrw)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        cin = int(weight.shape[1])
        if not dz.is_contiguous(memory_format=torch.channels_last):
            dz = dz.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        wino_dx = ctx.needs_input_grad[0] and ctx.wino_bwd
        if wino_dx and ctx.wino_bwd == 'c32':
            dx = _fe.conv3x3_c32(dz, weight, transposed=True)
        elif wino_dx:
            dx = _fe.conv3x3_wino(dz, _fe.wino_pack_weights_device(weight, transposed=True), None, cin, out_nhwc=True, relu=False)
        wino_dw = ctx.needs_input_grad[1] and ctx.wino_wrw
        if wino_dw:
            dw = _fe.conv3x3_wino_wrw(x, dz, like=weight)
        need = [ctx.needs_input_grad[0] and not wino_dx, ctx.needs_input_grad[1] and not wino_dw, False]
        if need[0] or need[1]:
            gi, gw, _ = torch.ops.aten.convolution_backward(dz, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, need)
            dx = gi if need[0] else dx
            dw = gw if need[1] else dw
        return dx, dw, None, None, None


def _wino_train_conv(conv: nn.Conv2d, x: torch.Tensor):
    """(forward by Winograd?, backward-data by Winograd?, weight gradient by Winograd?) for this layer and input, or None:
    MIOpen for everything."""
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    if not (WINO_TRAIN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and pair(conv.kernel_size) == (3, 3) and pair(conv.padding) == (1, 1) and pair(conv.stride) == (1, 1)
            and pair(conv.dilation) == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() < (1 << 30)
            and x.shape[0] * x.shape[2] * x.shape[3] * max(conv.in_channels, conv.out_channels) < (1 << 30)):
        return None
    ci, co, big = conv.in_channels, conv.out_channels, max(conv.in_channels, conv.out_channels)
    fwd = ci % 8 == 0 and co % 64 == 0 and big >= WINO_TRAIN_MIN_C_FWD
    bwd = co % 8 == 0 and ci % 64 == 0 and big >= WINO_TRAIN_MIN_C_BWD
    wrw = WINO_TRAIN_WRW and ci % 32 == 0 and co % 32 == 0 and x.shape[0] * x.shape[2] * x.shape[3] * big < (1 << 29)
    if C32_TRAIN and ci == 32 and co == 32:   # block 1's second layer: the inference engine's fp32-MFMA kernel, bare
        fwd = bwd = 'c32'
    return (fwd, bwd, wrw) if (fwd or bwd or wrw) else None


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
        if (FUSED_BN_RELU and self.training and x.is_cuda and len(self) == 3 and isinstance(self[1], nn.BatchNorm2d)
                and not torch.is_autocast_enabled()):
            conv, bn = self[0], self[1]
            if x.dtype == torch.float32 and conv.out_channels % 4 == 0 and bn.track_running_stats and bn.momentum is not None:
                if FUSED_CONV0 and pool is None and _is_first_layer_conv(conv, x):
                    _count_batch(bn)  # the model's first layer: convolution recomputed inside the passes
                    return _FusedConv0BNReLU.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                                   bn.running_var, bn.eps, bn.momentum)
                wino = _wino_train_conv(conv, x)
                if wino is not None:
                    z = _WinoConv3x3.apply(x, conv.weight, wino[0], wino[1], wino[2])
                else:
                    z = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
                if z.is_contiguous(memory_format=torch.channels_last):
                    _count_batch(bn)
                    fold = FUSED_BN_POOL and _is_pool_2x2_same(pool)
                    y = _FusedBiasBNReLU.apply(z, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                               bn.eps, bn.momentum, fold)
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
        if (FUSED_BN_RELU and FUSED_FC_BN and self.training and bn is not None and isinstance(self.act, nn.ReLU) and x.is_cuda
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
            if (FUSED_LSTM and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
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
        if x.is_cuda and ZERO_POOL:
            _ZERO_POOL.begin_step(x.device)
        y_pred = self._call(x)
        loss = self.loss_fn(y, y_pred)
        mark('forward')
        loss.backward()  # under DDP the bucketed RCCL all-reduce overlaps with this
        mark('backward')
        fused = self.use_agc and x.is_cuda  # one HIP launch for AGC + clipvalue over the whole model
        if fused:
            if self._fused_agc is None:
                object.__setattr__(self, '_fused_agc', FusedAGC(list(self.parameters())))
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


WINO_CONVS = os.environ.get("IRIS_WINO", "1") != "0"   # blocks 2-5 of the InferenceEngine as Winograd F(2x2, 3x3) on the fp32 MFMA


class _WinoStack(nn.Module):
    """Inference form of a run of ConvMPBlocks with 8 | Cin and 64 | Cout (blocks 2-5 of the CRNN, sj_train.py:222-242): every
    Conv2D 3x3 + folded bias + ReLU (+ the block's MaxPool) as ONE launch of the Winograd F(2x2, 3x3) kernel on the fp32
    matrix cores (iris_conv3x3_wino_bias_relu: 16 instead of 36 multiplies per output tile; MIOpen's implicit GEMMs already
    sit at the fp32 MFMA rate).  The layers hand each other the channel-chunked activation [B, C / 8, H, W, 8]; the first one
    converts from channels_last, the last one writes channels_last again."""

    def __init__(self, blocks):
        super().__init__()
        self.layers = []  # (index, cout, pool)
        dev = None
        for blk in blocks:
            convs = list(blk.convs)
            has_pool = isinstance(blk.pool, nn.MaxPool2d) or getattr(blk, '_pool_fused', False)
            for i, m in enumerate(convs):
                conv = m[0] if isinstance(m, nn.Sequential) else m
                w, b = conv.weight.detach(), conv.bias.detach()
                dev = w.device
                k = len(self.layers)
                self.register_buffer(f"packed{k}", _fe.wino_pack_weights(w), persistent=False)
                self.register_buffer(f"bias{k}", b.to(torch.float32).contiguous().clone(), persistent=False)
                self.layers.append((k, int(w.shape[0]), has_pool and i == len(convs) - 1))

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
        for k, cout, pool in self.layers:
            x = _fe.conv3x3_wino_bias_relu(x, getattr(self, f"packed{k}"), getattr(self, f"bias{k}"), cout, pool=pool, out_nhwc=(k == last))
        return x


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


FUSED_LSTM = os.environ.get("IRIS_FUSED_LSTM", "1") != "0"


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
        if fuse_epilogues and hip_convs and WINO_CONVS and dev.type == 'cuda':
            # the trailing run of plain ConvMPBlocks whose convolutions the Winograd kernel takes (blocks 2-5 of v9): one module
            feats = list(self.model.features)
            k = len(feats)
            while k > 1 and _WinoStack.eligible(feats[k - 1]):
                k -= 1
            if k < len(feats):
                stack = _WinoStack(feats[k:])
                self.wino_convs = len(stack.layers)
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
    return torch.mean(-(y_true * torch.log(p) + (1 - y_true) * torch.log(1 - p)))


def make_optimizer(config, params, capturable: bool = False):
    """`capturable`: Adam with its step count and learning rate in device tensors, for `GraphedTrainStep`."""
    params = list(params)
    if capturable:
        if config.optimizer != 'adam' or not (params and params[0].is_cuda):
            raise ValueError("make_optimizer(capturable=True): Adam on a GPU")
        return torch.optim.Adam(params, lr=torch.tensor(float(config.lr), device=params[0].device), eps=1e-7, fused=True,
                                capturable=True)
    # foreach=True on a GPU: the update AND zero_grad run as a handful of multi-tensor kernels instead of one per
    # parameter (86 fills of ~3.6 us each per step otherwise)
    fe = bool(params) and params[0].is_cuda
    if config.optimizer == 'adam':
        if fe and os.environ.get("IRIS_ADAM_FUSED", "1") != "0":  # the whole update in one multi-tensor launch
            return torch.optim.Adam(params, lr=config.lr, eps=1e-7, fused=True)
        return torch.optim.Adam(params, lr=config.lr, eps=1e-7, foreach=fe)  # Keras Adam epsilon
    if config.optimizer == 'sgd':
        return torch.optim.SGD(params, lr=config.lr, momentum=0.9, foreach=fe)
    if config.optimizer == 'rmsprop':
        return torch.optim.RMSprop(params, lr=config.lr, momentum=0.9, alpha=0.9, eps=1e-7, foreach=fe)
    raise ValueError('adabelief is deprecated')


def run_name(config) -> str:
    """Run name encoding of sj_train.py:416-429."""
    name = (config.name + '_') if config.name != '' else ''
    first = {'eff': f'B{config.model}', 'se': 'se', 'vad': 'vad'}[config.model_type]
    name += '_'.join([first, f'v{config.v}', f'lr{config.lr}', f'batch{config.batch_size}',
                      f'opt_{config.optimizer}', f'mel{config.n_mels}', f'chan{config.n_chan}',
                      f'{config.loss.upper()}', f'framelen{config.n_frame}'])
    return name if name.endswith('.h5') else name + '.h5'


def configure_miopen() -> None:
    """MIOpen defaults for this model's fp32 conv shapes (only set when the user has not):
    NORMAL find benchmarks the applicable solvers once per shape - the FAST heuristic picks a CK
    backward-weight kernel that is ~60x slower here - and the naive reference solvers (hundreds of
    ms per call, never the winner) are kept out of that benchmark."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")
    for d in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + d, "0")
    _use_shipped_miopen_db()


def _use_shipped_miopen_db() -> None:
    """challenge_amd/miopen_db/ holds MIOpen's user perf-db / find-db after an exhaustive search (MIOPEN_FIND_ENFORCE=3,
    scripts/gpu_miopen_tune.sh) over this model's convolution shapes at batch 64 on an MI355X: tuned kernel parameters for
    the solvers MIOpen already has, 16.2 -> 15.1 ms per training step (profiles/r3/miopen_tune.log).  MIOpen also WRITES to
    its user db, so a per-user, per-rank copy (miopen_db/_run/, named after the shipped content) is what MIOPEN_USER_DB_PATH points at.
    Skipped when the user has set MIOPEN_USER_DB_PATH, or with IRIS_MIOPEN_DB=0; the files are keyed by MIOpen build and
    GPU, so any other build / GPU simply does not find them."""
    if "MIOPEN_USER_DB_PATH" in os.environ or os.environ.get("IRIS_MIOPEN_DB", "1") == "0":
        return
    import hashlib
    import shutil
    import tempfile
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
    try:
        files = sorted(f for f in os.listdir(src) if f.endswith("db.txt"))
        if not files:
            return
        digest = hashlib.sha256()
        for f in files:
            with open(os.path.join(src, f), "rb") as fh:
                digest.update(f.encode() + b"\0" + fh.read())
        uid = os.getuid() if hasattr(os, "getuid") else 0
        name = f"u{uid}_r{os.environ.get('LOCAL_RANK', '0')}_{digest.hexdigest()[:12]}"
        dst = os.path.join(src, "_run", name)  # beside the shipped files (git- and gpurun-ignored) ...
        try:
            os.makedirs(dst, exist_ok=True)
        except OSError:  # ... or, for a read-only installation, in the temp dir
            dst = os.path.join(tempfile.gettempdir(), "iris_miopen_db_" + name)
            os.makedirs(dst, exist_ok=True)
        for f in files:
            if not os.path.exists(os.path.join(dst, f)):
                tmp = os.path.join(dst, f + f".{os.getpid()}.tmp")
                shutil.copyfile(os.path.join(src, f), tmp)
                os.replace(tmp, os.path.join(dst, f))
        os.environ["MIOPEN_USER_DB_PATH"] = dst
    except OSError:
        pass  # no shipped db / unwritable temp dir: MIOpen's own defaults


def distributed_env(env=None):
    """Environment a multi-process GPU job needs on this stack, set BEFORE the first GPU call of the process (or in the
    environment handed to the ranks): the host driver only supports dmabuf IPC, and without HSA_ENABLE_IPC_MODE_LEGACY=0
    RCCL's intra-node transports fail with `hipIpcGetMemHandle: invalid argument`.  `bench.self_launch`, `init_distributed`
    and INTEGRATION.md's launch line all go through here, so the three cannot drift apart.  Values the user has set win."""
    env = os.environ if env is None else env
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')  # single node; the container's hostname may not resolve
    return env


def miopen_db_status() -> str:
    """Did MIOpen pick up the shipped perf-db / find-db (`_use_shipped_miopen_db`)?  Call AFTER the model's convolutions have
    run once.  The files are keyed by MIOpen's build string and the GPU (arch + CU count) in their NAMES: a matching MIOpen
    reads and appends to the shipped names, any other build ignores them and - having had to search - writes files under its
    own name next to them.  'used' / 'ignored: ...' / 'off: ...' (bench.py records it as extra.miopen_db: the tuned db is
    worth 13.7 vs 15.1 ms per training step, so a line must say which of the two it measured)."""
    path = os.environ.get("MIOPEN_USER_DB_PATH")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
    if os.environ.get("IRIS_MIOPEN_DB", "1") == "0":
        return "off: IRIS_MIOPEN_DB=0"
    try:
        shipped = {f for f in os.listdir(src) if f.endswith("db.txt")}
    except OSError:
        shipped = set()
    if not path or not shipped:
        return "off: no shipped db in use"
    if not (os.path.basename(os.path.dirname(path)) == "_run" or os.path.basename(path).startswith("iris_miopen_db_")):
        return "off: MIOPEN_USER_DB_PATH set by the user"
    try:
        present = {f for f in os.listdir(path) if f.endswith("db.txt")}
    except OSError:
        return "off: " + path + " unreadable"
    foreign = sorted(present - shipped)
    if foreign:
        return ("ignored: this MIOpen build wrote " + ", ".join(foreign[:2]) + " - its build string / GPU differs from the shipped "
                + sorted(shipped)[-1])
    return "used"


def force_process_group() -> bool:
    """IRIS_FORCE_PG=1: create the process group, wrap the model in DistributedDataParallel and run every collective of
    `fit` even at world size 1.  A one-GPU box then executes the REAL backend (RCCL: communicator initialisation under
    HSA_ENABLE_IPC_MODE_LEGACY=0, DDP's reducer on RCCL's stream next to the raw-pointer HIP passes on torch's current
    stream, the all-reduce kernels themselves) - what a gloo run with two ranks sharing the device cannot show."""
    return os.environ.get('IRIS_FORCE_PG', '0') == '1'


def collectives_on(world: int) -> bool:
    """Do `fit` / `average_bn_statistics` / the bench issue their collectives?  world > 1, or a forced group at world 1."""
    return world > 1 or (force_process_group() and torch.distributed.is_available() and torch.distributed.is_initialized())


def init_distributed(force_group: Optional[bool] = None):
    """One process per GPU (torchrun): returns (rank, world, device).  Backend 'nccl' is
    RCCL on ROCm; 'gloo' on CPU-only hosts (tests).  `force_group` (default: IRIS_FORCE_PG=1) creates the group at world
    size 1 as well - in a fresh process, at its first GPU call, never after a re-exec."""
    distributed_env()  # before torch.cuda.is_available(): that call already initialises the HIP runtime
    if force_group:
        os.environ['IRIS_FORCE_PG'] = '1'
    force_group = force_process_group() if force_group is None else force_group
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    else:
        device = torch.device('cpu')
    if (world > 1 or force_group) and not torch.distributed.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if device.type == 'cuda':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    return rank, world, device


# Gradient buckets of the v9 CRNN (39.5 MB, filled in reverse layer order): with 25 MB the last bucket is 22 MB (block 4's
# first two convolutions and everything below) and its all-reduce starts only when backward has finished, fully exposed; with
# 12 MB the buckets are 0.9 / 7.2 / 9.4 / 9.4 / 11.8 / 0.7 MB - the 11.8 MB one goes out while blocks 2 and 1 (40 % of backward)
# still compute, and what is left after backward is 0.7 MB.
DDP_BUCKET_MB = int(os.environ.get("IRIS_DDP_BUCKET_MB", "12"))


def wrap_ddp(model: CustomModel, device, world: int):
    if not collectives_on(world):
        return None
    from torch.nn.parallel import DistributedDataParallel as DDP
    # ONE collective per step - the bucketed gradient all-reduce: BatchNorm statistics stay per replica during the epoch
    # (the reference has no multi-GPU at all), so the per-forward buffer broadcast is switched off; `fit` averages them
    # over the ranks once per epoch, before validation and checkpointing (average_bn_statistics)
    return DDP(model, device_ids=[device.index] if device.type == 'cuda' else None,
               bucket_cap_mb=DDP_BUCKET_MB, gradient_as_bucket_view=True, broadcast_buffers=False)


@torch.no_grad()
def average_bn_statistics(model: nn.Module, world: int) -> None:
    """BatchNorm running statistics are per replica under DDP (`broadcast_buffers=False`: no per-forward broadcast), each
    rank seeing 1 / world of the data.  Before validation and checkpointing they are averaged over the ranks - ONE small
    all-reduce per epoch over the 46 running_mean / running_var vectors flattened together - so that every rank validates,
    and rank 0 saves, the same model.  (The mean of per-rank variances ignores the spread of the per-rank means: the
    running averages of identically distributed shards, where that spread is O(1 / sqrt(steps)).)"""
    if not collectives_on(world):
        return
    bufs = [b for name, b in model.named_buffers() if name.endswith(('running_mean', 'running_var'))]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    torch.distributed.all_reduce(flat)
    flat /= world
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b))
        off += n
    if hasattr(model, 'bump_generation'):
        model.bump_generation()


class GraphedTrainStep:
    """`model.train_step` (forward, loss, backward, AGC + clipvalue, optimiser) as ONE replayed hipGraph - single GPU, fixed
    batch shape.  The gradients live in the graph's memory pool and are dropped inside the capture, so the replay has neither
    the zero fills nor autograd's accumulate launches; 13.77 -> 13.55 ms per batch-64 step on an MI355X.

        opt = make_optimizer(config, model.parameters(), capturable=True)      # learning rate held in a device tensor
        model.compile(opt, loss, clipvalue=...)
        step = GraphedTrainStep(model, (x, y))                                   # 3 eager warm-up steps (they train), then the capture
                                                                                 # (preserve_state=True: the warm-up leaves no trace)
        for x, y in data: loss = step((x, y))['loss']                            # inputs are copied into the static buffers
        step.set_lr(value)                                                       # schedulers write the tensor

    MIOpen must already know its kernels for these shapes (the warm-up steps see to that).  Not under DDP."""

    def __init__(self, model: "CustomModel", example, warmup: int = 3, preserve_state: bool = False):
        """`preserve_state`: parameters, buffers and the optimiser's state are put back after the warm-up steps, so that the
        first replay is the FIRST update the example batch causes (what `fit` wants: one update per batch, as the reference)."""
        x, y = example
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep: a GPU tensor is required (hipGraph capture; no CPU fallback)")
        if model._ddp is not None:
            raise RuntimeError("GraphedTrainStep: not under DistributedDataParallel")
        opt = model.optimizer
        if not all(g.get('capturable', False) for g in opt.param_groups):
            raise ValueError("GraphedTrainStep: the optimiser must be capturable - make_optimizer(config, params, capturable=True)")
        self.model, self.x, self.y = model, x.clone(), y.clone()
        saved = None
        if preserve_state:
            tensors = list(model.parameters()) + list(model.buffers())
            saved = ([t.detach().clone() for t in tensors], tensors,
                     {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in opt.state.get(p, {}).items()}
                      for g in opt.param_groups for p in g['params']})
        # `warmup` eager steps in all (they train the model): all but the last on the current stream - without them the
        # capture was invalidated on this stack (some first-use initialisation that a side stream alone does not trigger) -
        # and the last one on a side stream, as torch's capture recipe asks
        for _ in range(max(warmup, 2) - 1):
            model.train_step((self.x, self.y))
        torch.cuda.synchronize(x.device)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            model.train_step((self.x, self.y))
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        object.__setattr__(model, '_fused_agc', None)  # its table holds the eager gradients' addresses
        opt.zero_grad(set_to_none=True)
        if saved is not None:   # undo the warm-up in place (the graph will be captured on these very tensors)
            with torch.no_grad():
                for t, old in zip(saved[1], saved[0]):
                    t.copy_(old)
                for g in opt.param_groups:
                    for p in g['params']:
                        before, now = saved[2][id(p)], opt.state.get(p, {})
                        for k, v in now.items():
                            if torch.is_tensor(v):   # moments and the step count: back to their old values, or to a fresh 0
                                v.copy_(before[k]) if k in before else v.zero_()
            model.bump_generation()
            torch.cuda.synchronize(x.device)
        # The captured AGC launch reads a table whose pinned staging buffer is the source of a captured copy node: this
        # object owns both for as long as the graph lives, and the model's own `_fused_agc` stays None - an eager
        # `model.train_step` later (e.g. a ragged last batch) builds a SEPARATE FusedAGC instead of rebuilding - and
        # freeing - the buffers the graph replays from.
        self._agc = FusedAGC(list(model.parameters())) if model.use_agc else None
        if self._agc is not None:
            self._agc.reserve()
            torch.cuda.synchronize(x.device)
        self.graph = torch.cuda.CUDAGraph()
        # (thread_local: another thread's harmless queries - RCCL's watchdog polling the events of earlier collectives when a
        # process group is alive in this process - must not invalidate the capture, nor be killed by it)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            model.train()
            if ZERO_POOL:
                _ZERO_POOL.begin_step(x.device)
            loss = model.loss_fn(self.y, model._call(self.x))
            loss.backward()
            if model.use_agc:
                self._agc(0.01, 1e-3, model.clipvalue)
                self._agc.freeze()
            elif model.clipvalue:
                torch.nn.utils.clip_grad_value_([p for p in model.parameters() if p.grad is not None], model.clipvalue)
            opt.step()
            opt.zero_grad(set_to_none=True)
            self.loss = loss.detach()
        torch.cuda.synchronize(x.device)

    def __call__(self, data):
        x, y = data
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        self.model.bump_generation()  # a replay moves parameters and BatchNorm statistics behind ATen's back
        return {'loss': self.loss}

    def set_lr(self, value: float) -> None:
        for g in self.model.optimizer.param_groups:
            if torch.is_tensor(g['lr']):
                g['lr'].fill_(float(value))
            else:
                g['lr'] = float(value)


_PLAN_CHECK_ON_CPU = False  # test hook (tests/test_ddp_gloo.py): consult the frontend plans' status for a CPU-resident loss too


# fit / main run the training step as ONE replayed hipGraph wherever that is possible (one GPU without DDP, Adam, batches of one
# shape): the step then costs what its kernels cost however slow the host is at launching ~260 of them.  IRIS_GRAPH_STEP=0: eager.
GRAPH_STEP = os.environ.get("IRIS_GRAPH_STEP", "1") != "0"


def fit(model: CustomModel, train_set, epochs, steps_per_epoch, validation_data=None, validation_steps=16,
        scheduler=None, csv_path=None, checkpoint_path=None, patience=None, rank=0, world=1, verbose=True,
        swa=None, graph: Optional[bool] = None):
    """Minimal Keras-fit equivalent for this path: per-epoch LR schedule, CSV log,
    best-val-loss checkpoint, early stopping, TerminateOnNaN (sj_train.py:489-519).
    `graph` (default: on, IRIS_GRAPH_STEP=0 switches it off): run the training step as ONE replayed hipGraph (GraphedTrainStep) - single GPU
    without DDP, a capturable optimiser (make_optimizer(..., capturable=True)), batches of one shape; a batch of another
    shape (a ragged last one) takes the eager step.  The step then costs what its kernels cost (10.05 ms per batch of 64)
    however slow the host is at launching ~260 kernels."""
    best, bad, history = math.inf, 0, []
    coll = collectives_on(world)  # world > 1, or a forced process group at world 1 (IRIS_FORCE_PG=1)
    it = iter(train_set)
    graph = GRAPH_STEP if graph is None else bool(graph)
    graph = graph and model._ddp is None and all(g.get('capturable', False) for g in model.optimizer.param_groups)
    gstep = None

    def one_step(data):
        nonlocal gstep
        x, y = data
        if not (graph and x.is_cuda):
            return model.train_step(data)
        if gstep is None:
            try:   # the warm-up steps run on this batch and are undone: the first replay is its first update
                gstep = GraphedTrainStep(model, data, preserve_state=True)
            except Exception as exc:   # a capture that fails must not take the training run down: eager from here on
                if verbose and rank == 0:
                    print(f"fit: the training step could not be captured as a hipGraph ({exc!r:.200}); running it eagerly")
                gstep = False
        if gstep and x.shape == gstep.x.shape and y.shape == gstep.y.shape and x.dtype == gstep.x.dtype:
            return gstep(data)
        return model.train_step(data)

    for epoch in range(epochs):
        if scheduler is not None:
            lr = scheduler(epoch)
            for g in model.optimizer.param_groups:
                if torch.is_tensor(g['lr']):
                    g['lr'].fill_(float(lr))   # capturable optimiser: the rate lives in a device tensor (a graph reads it)
                else:
                    g['lr'] = lr
        t0, losses = time.time(), []
        for _ in range(steps_per_epoch):
            losses.append(one_step(next(it))['loss'].clone() if graph else one_step(next(it))['loss'])
        loss = torch.stack(losses).mean()
        # The one place per epoch where the frontend plans' status words are read for certain (the hot path also reports a
        # failed earlier launch at the plan's next call, without a sync): EpilogueTimeout naming the plan instead of training
        # on NaN features.  Under DDP the failure of ONE rank must not leave the others waiting in the collectives below, so
        # the verdict rides along with the epoch loss in the same all-reduce and every rank raises after it.
        plan_failure = None
        if loss.is_cuda or _PLAN_CHECK_ON_CPU:
            try:
                _fe.check_plans(loss.device)
            except _fe.N.EpilogueTimeout as exc:
                plan_failure = exc
        if coll:
            pack = torch.stack([loss, loss.new_tensor(1.0 if plan_failure is not None else 0.0)])
            torch.distributed.all_reduce(pack)  # two scalars per epoch
            loss, failed_ranks = pack[0] / world, int(round(float(pack[1])))
            if failed_ranks and plan_failure is None:
                plan_failure = _fe.N.EpilogueTimeout(f"{failed_ranks} other rank(s) of this job reported a failed fused min-max / "
                                                     "log epilogue (NaN features); stopping with them")
        if plan_failure is not None:
            raise plan_failure
        row = {'epoch': epoch, 'loss': float(loss), 'lr': float(model.optimizer.param_groups[0]['lr']),
               'time': time.time() - t0}
        if coll:
            average_bn_statistics(model, world)
        if not math.isfinite(row['loss']):
            if verbose and rank == 0:
                print('NaN loss, terminating')
            break
        if validation_data is not None:
            vit = iter(validation_data)
            vl = torch.stack([model.test_step(next(vit))['loss'] for _ in range(validation_steps)]).mean()
            if coll:  # every rank validates its own shard: the monitored value is the mean over ranks
                torch.distributed.all_reduce(vl)
                vl = vl / world
            row['val_loss'] = float(vl)
        history.append(row)
        if swa is not None:
            swa.on_epoch_end(epoch, model)
        # The monitored value is identical on every rank (all-reduced above), so best / bad / stop are
        # computed by all ranks alike and they leave the loop together; only file I/O is rank 0's.
        monitor = row.get('val_loss', row['loss'])
        improved = monitor < best
        if improved:
            best, bad = monitor, 0
        else:
            bad += 1
        if rank == 0:
            if verbose:
                print(row)
            if csv_path:
                new = not os.path.exists(csv_path)
                with open(csv_path, 'a', newline='') as f:
                    w = csv.DictWriter(f, fieldnames=list(row))
                    if new:
                        w.writeheader()
                    w.writerow(row)
            if improved and checkpoint_path:
                torch.save(model.state_dict(), checkpoint_path)
        stop = patience is not None and not improved and bad >= patience  # Keras EarlyStopping: wait >= patience, tested on a non-improving epoch
        if coll:  # belt and braces: one int per epoch, rank 0's decision wins
            flag = torch.tensor([1 if stop else 0], dtype=torch.int32, device=loss.device)
            torch.distributed.broadcast(flag, src=0)
            stop = bool(int(flag.item()))
        if stop:
            break
    return history


def main(argv=None):
    config = ARGS().get(argv)
    config.loss = config.loss.upper()
    if config.loss != 'MSE':
        config.mse_multiplier = 1
    configure_miopen()
    rank, world, device = init_distributed()
    if rank == 0:
        print(config)
    NAME = run_name(config)
    model = get_model(config).to(device).to(memory_format=torch.channels_last)
    # one GPU, Adam (and not IRIS_GRAPH_STEP=0): the step as one replayed hipGraph - the optimiser then keeps its rate on the device
    opt = make_optimizer(config, model.parameters(),
                         capturable=GRAPH_STEP and world == 1 and device.type == 'cuda' and config.optimizer == 'adam')
    loss = binary_crossentropy if config.loss == 'BCE' else \
        (lambda yt, yp: sigmoid_focal_crossentropy(yt, yp).mean())
    model.compile(opt, loss, clipvalue=None if config.no_clipvalue_after_agc else config.clipvalue,
                  ddp=wrap_ddp(model, device, world))
    if rank == 0:
        print(NAME, sum(p.numel() for p in model.parameters()), 'parameters')
    if config.pretrain:
        # `model.load_weights(NAME)` (sj_train.py:467-469): this module's own .pt checkpoint, or - a model trained with the
        # reference - its Keras weights as an .npz next to it (scripts/dump_keras_weights.py writes one from the .h5)
        if os.path.exists(NAME.replace('.h5', '.pt')):
            model.load_state_dict(torch.load(NAME.replace('.h5', '.pt'), map_location=device))
            if rank == 0:
                print('loaded pretrained model', NAME.replace('.h5', '.pt'))
        elif os.path.exists(NAME.replace('.h5', '.npz')):
            load_keras_weights(model, NAME.replace('.h5', '.npz'))
            if rank == 0:
                print('loaded pretrained Keras weights', NAME.replace('.h5', '.npz'))
    if device.type == 'cuda' and config.online_stft:
        # corpora resident in HBM as WAVEFORMS, mixed before the STFT, fused frontend on line (synthetic sources:
        # the reference's pickles hold spectra, not waveforms)
        dd = not config.host_draws
        train_set = make_wave_dataset(config, training=True, device=device, seed=1000 + rank, device_draw=dd)
        test_set = make_wave_dataset(config, training=False, device=device, seed=2000 + rank, device_draw=dd)
    elif device.type == 'cuda' and not config.per_sample_pipeline:
        # corpora resident in HBM, whole batches synthesised on the device (each rank draws its own stream)
        dd = not config.host_draws
        train_set = make_device_dataset(config, training=True, device=device, seed=1000 + rank, device_draw=dd)
        test_set = make_device_dataset(config, training=False, device=device, seed=2000 + rank, device_draw=dd)
    else:
        train_set = make_dataset(config, training=True)
        test_set = make_dataset(config, training=False)
    from .swa import NO_SWA_ERROR, SWA
    swa = SWA(start_epoch=config.epochs // 4, swa_freq=2)  # sj_train.py:491
    fit(model, train_set, config.epochs, config.steps_per_epoch, test_set, config.validation_steps,
        scheduler=custom_scheduler(4096, config.epochs / 12, config.lr_div),
        csv_path=NAME.replace('.h5', '.csv'), checkpoint_path=NAME.replace('.h5', '.pt'),
        patience=config.patience, rank=rank, world=world, swa=swa)
    try:
        swa.finalize(model)
        if rank == 0:
            torch.save(model.state_dict(), NAME.replace('.h5', '_SWA.pt'))
            print('best model:', NAME.replace('.h5', '_SWA.pt'))
    except NO_SWA_ERROR:
        pass
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(NAME.split('.h5')[0])


if __name__ == "__main__":
    main()
