"""Drop-in counterpart of the on-path part of the reference's trainer.py (the older
driver): `minmax_log_on_mel` (trainer.py:63-77), `augment` (:80-83), `preprocess_labels`
(:86-94), `to_density_labels` (:97-104), `make_dataset` (:107-141), `cos_sim`
(:192-198), `custom_scheduler` (:201-210).  Its EfficientNet density-regression model
and `custom_loss` (:144-189, :213-289) are outside the accelerated path."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from .data_utils import augment, minmax_log_on_mel  # noqa: F401  (same definitions as trainer.py:63-83)
from .dataset import AUTOTUNE
from .pipeline import make_pipeline
from .sj_train import complex_to_mel, custom_scheduler, synthetic_sources  # noqa: F401
from .utils import load_data, safe_div

args = argparse.ArgumentParser()
args.add_argument('--name', type=str, default='')
args.add_argument('--model', type=str, default='EfficientNetB4')
args.add_argument('--n_chan', type=int, default=1)
args.add_argument('--n_classes', type=int, default=3)
args.add_argument('--datapath', type=str, default='/root/datasets/Interspeech2020/generate_wavs/codes')
args.add_argument('--background_sounds', type=str, default='drone_normed_complex_v3.pickle')
args.add_argument('--voices', type=str, default='voice_normed_complex_v3.pickle')
args.add_argument('--labels', type=str, default='voice_labels_mfc_v3.npy')
args.add_argument('--noises', type=str, default='noises_specs_v2.pickle')
args.add_argument('--test_background_sounds', type=str, default='dummy_specs.pickle')
args.add_argument('--test_voices', type=str, default='dummy_specs.pickle')
args.add_argument('--test_labels', type=str, default='dummy_labels.npy')
args.add_argument('--n_mels', type=int, default=80)
args.add_argument('--batch_size', type=int, default=12)
args.add_argument('--n_frame', type=int, default=2048)
args.add_argument('--multiplier', type=float, default=10)
args.add_argument('--snr', type=float, default=-15)
args.add_argument('--max_voices', type=int, default=10)
args.add_argument('--max_noises', type=int, default=6)


def preprocess_labels(multiplier):
    """Five sum-pool-by-2 passes ('SAME': a ragged tail is averaged over its valid entries
    and doubled, as tf.nn.avg_pool1d(...)*2 does) then * multiplier (trainer.py:86-94)."""
    def _preprocess(x, y):
        for _ in range(5):
            yt = y.transpose(1, 2)
            yt = torch.nn.functional.avg_pool1d(yt, 2, 2, ceil_mode=True, count_include_pad=False) * 2
            y = yt.transpose(1, 2)
        y = y * multiplier
        return x, y
    return _preprocess


def to_density_labels(x, y):
    """[..., n_voices, n_frames, n_classes] -> per-voice unit mass, summed over voices
    (trainer.py:97-104)."""
    y = safe_div(y, torch.sum(y, dim=(-2, -1), keepdim=True))
    y = torch.sum(y, dim=-3)
    return x, y


def make_dataset(config, training=True, n_classes=3, sources=None):
    """Stage order of trainer.py:107-141."""
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
    labels = np.eye(n_classes, dtype='float32')[np.asarray(labels)]
    pipeline = make_pipeline(backgrounds, voices, labels, noises, n_frame=config.n_frame,
                             max_voices=config.max_voices, max_noises=config.max_noises, n_classes=n_classes,
                             snr=config.snr, min_ratio=1)
    pipeline = pipeline.map(to_density_labels)
    if training:
        pipeline = pipeline.map(augment)
    pipeline = pipeline.batch(config.batch_size, drop_remainder=False)
    pipeline = pipeline.map(complex_to_mel(config.n_mels, int(np.asarray(backgrounds[0]).shape[0])))
    pipeline = pipeline.map(minmax_log_on_mel)
    pipeline = pipeline.map(preprocess_labels(config.multiplier))
    return pipeline.prefetch(AUTOTUNE)


def cos_sim(y_true, y_pred):
    """trainer.py:192-198: cosine similarity over the time axis, averaged over the classes
    that are present."""
    m = (torch.sum(y_true, dim=-2) > 0.).to(torch.float32)
    m = safe_div(m, torch.sum(m, dim=-1, keepdim=True))
    cs = -torch.nn.functional.cosine_similarity(y_true, y_pred, dim=-2, eps=1e-12)  # Keras returns the negative
    return torch.sum(cs * m, dim=-1)
