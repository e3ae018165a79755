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
        self._seed, self._ddraw = int(seed), None

    def draw_bands(self, batch: int, n_time: int):
        """Host draw (NumPy Generator): exact integer distributions of transforms.py:25-26."""
        return _du.augment_draw_batch(batch, n_time, self.plan.n_bins, self.rng)

    def draw_bands_device(self, batch: int, n_time: int):
        """Device draw, no host round trip, ONE launch (`iris_augment_draw`: Philox keyed by the seed, call counter in device
        memory; the exact integer distributions of transforms.py:25-26; capturable into a hipGraph).  Returns long-lived int32
        device tensors (t_bands [B, 6, 2], f_bands [B, 1, 2]) that the next draw of the same shape overwrites.  (Until round 6
        this was ~20 small torch launches per call - randint, rand, floor, minimum, stack ... - a tenth of a millisecond in front
        of every training step: scripts/gpu_op_census.py.)"""
        if self._ddraw is None:
            self._ddraw = _du.DeviceAugmentDraw(self.plan.device, seed=self._seed)
        return self._ddraw(batch, n_time, self.plan.n_bins)

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


# ---------------------------------------------------------------------------
# the rest of the reference's sj_train surface lives in sibling modules (round 6 split) and is re-exported under the
# reference's names: model.py (define_keras_model, get_model, CustomModel, ConvMPBlock, FullyConnectedLayer :191-255, :402-403),
# hip_autograd.py (adaptive_clip_grad :145-155 and the HIP passes), distributed.py (absent upstream), fit.py (:434-519)
# ---------------------------------------------------------------------------
from . import switches as SW  # noqa: E402
from .hip_autograd import (FusedAGC, _BiLSTM128, _FusedBiasBNReLU, _FusedConv0BNReLU, _IN_STEP, _WinoConv3x3, _ZERO_POOL,  # noqa: E402,F401
                           _ZeroPool, _is_first_layer_conv, _is_pool_2x2_same, _lstm_is_bilstm128, _wino_train_conv, _zeros,
                           adaptive_clip_grad, bilstm128)
from .model import (ConvMPBlock, CustomModel, FullyConnectedLayer, InferenceEngine, _Bottleneck, _ConvBNReLU,  # noqa: E402,F401
                    _ConvBiasReLU, _HipBiLSTM, _SmoothPool, _WinoStack, _keras_weight_list, binary_crossentropy,
                    define_keras_model, fold_batchnorm, get_model, keras_weight_shapes, load_keras_weights)
from .distributed import (_gradient_buckets, average_bn_statistics, collectives_on, distributed_env,  # noqa: E402,F401
                          force_process_group, init_distributed, wrap_ddp)
from .fit import (GraphCaptureError, GraphedTrainStep, configure_miopen, fit, graph_step_possible, make_optimizer,  # noqa: E402,F401
                  miopen_db_status, run_name)


class _SjTrainModule(type(os)):
    """`sj_train.<SWITCH>` reads and writes the attribute of switches.py (one table, read at call time by every module), so
    `sj_train.WINO_TRAIN = False` / `monkeypatch.setattr(sj_train, "FUSED_BN_RELU", False)` keep working after the split."""

    def __getattr__(self, name):
        if name in SW.NAMES:
            return getattr(SW, name)
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in SW.NAMES:
            setattr(SW, name, value)
        else:
            super().__setattr__(name, value)


import sys as _sys  # noqa: E402
_sys.modules[__name__].__class__ = _SjTrainModule


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
    # a GPU, Adam (and not IRIS_GRAPH_STEP=0): the step as one replayed hipGraph - the optimiser then keeps its rate on the
    # device; with more than one rank the gradient all-reduce over RCCL is part of the graph (GraphedTrainStep)
    opt = make_optimizer(config, model.parameters(),
                         capturable=SW.GRAPH_STEP and device.type == 'cuda' and config.optimizer == 'adam')
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