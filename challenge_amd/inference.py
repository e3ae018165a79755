"""Sliding-window inference chain of the reference's `metrics.evaluate` (metrics.py:40-81)
up to the thresholded frame predictions -- the one place where the reference runs
STFT -> mel -> forward in a single call.  Scoring (event extraction, error rate) stays in
the reference's untouched metrics.py.

    wav -> load_wav (normalize + STFT, HIP) -> channel transform -> stft_filter(16)
        -> |.| -> mel -> minmax (per mel row: the tensor is unbatched here) -> log  (HIP)
        -> frames of n_frame with hop `overlap_hop`, zero-padded at the end
        -> model -> (upsample) -> overlap-add average -> avg-pool / max-pool smoothing -> >= 0.5
"""
from __future__ import annotations

import torch

from . import data_utils as D
from . import transforms as T
from .utils import label_downsample_model


def frame(x: torch.Tensor, frame_length: int, frame_step: int, pad_end: bool = True, axis: int = -2) -> torch.Tensor:
    """tf.signal.frame(x, frame_length, frame_step, pad_end, axis): the framed axis becomes
    [num_frames, frame_length]; with pad_end num_frames = ceil(len / step), zero padded."""
    axis = axis % x.dim()
    n = x.shape[axis]
    if pad_end:
        num = -(-n // frame_step)
        need = (num - 1) * frame_step + frame_length
        if need > n:
            pad = [0, 0] * x.dim()
            pad[2 * (x.dim() - 1 - axis) + 1] = need - n
            x = torch.nn.functional.pad(x, pad)
    else:
        num = max(0, 1 + (n - frame_length) // frame_step)
    idx = (torch.arange(num, device=x.device)[:, None] * frame_step
           + torch.arange(frame_length, device=x.device)[None, :])
    out = x.index_select(axis, idx.reshape(-1))
    return out.reshape(*x.shape[:axis], num, frame_length, *x.shape[axis + 1:])


def overlap_and_add(frames: torch.Tensor, frame_step: int) -> torch.Tensor:
    """tf.signal.overlap_and_add on [..., W, L] -> [..., (W-1)*step + L]."""
    w, length = frames.shape[-2], frames.shape[-1]
    out_len = (w - 1) * frame_step + length
    out = frames.new_zeros(*frames.shape[:-2], out_len)
    idx = (torch.arange(w, device=frames.device)[:, None] * frame_step
           + torch.arange(length, device=frames.device)[None, :]).reshape(-1)
    out.index_add_(-1, idx, frames.reshape(*frames.shape[:-2], w * length))
    return out


def _pool_same(x: torch.Tensor, k: int, mode: str) -> torch.Tensor:
    """Keras {Average,Max}Pooling1D(k, 1, padding='same') on [T, K]."""
    left = (k - 1) // 2
    right = k - 1 - left
    xt = x.t()[None]  # [1, K, T]
    F = torch.nn.functional
    if mode == 'max':
        return F.max_pool1d(F.pad(xt, (left, right), value=float('-inf')), k, 1)[0].t()
    ones = torch.ones((1, 1, x.shape[0]), dtype=x.dtype, device=x.device)
    cnt = F.avg_pool1d(F.pad(ones, (left, right)), k, 1)
    return (F.avg_pool1d(F.pad(xt, (left, right)), k, 1) / cnt)[0].t()


def smooth(preds: torch.Tensor, sr: int = 16000, hop: int = 256) -> torch.Tensor:
    """metrics.py:76-80: 0.5 s average pooling, then 2 s max pooling, stride 1, 'same'."""
    k = int(0.5 * sr) // hop
    return _pool_same(_pool_same(preds, k, 'avg'), 4 * k, 'max')


def features_for_eval(spec: torch.Tensor, config) -> torch.Tensor:
    """[F, T, 2C] complex spectrogram -> log-mel [M, T, C'] as metrics.py:42-54 prepares it."""
    inputs = spec
    if config.n_chan == 1:
        inputs = D.mono_chan(inputs)
    elif config.n_chan == 3:
        inputs = D.stereo_mono(inputs)
    elif config.n_chan > 3:
        inputs = D.random_merge_aug(config.n_chan)(inputs, None)
    inputs = D.stft_filter(int(round(256 * 1000 / 16000)))(inputs)
    inputs = T.complex_to_magphase(inputs)
    inputs = T.magphase_to_mel(config.n_mels, inputs.shape[0])(inputs)
    inputs = D.minmax(inputs)       # unbatched: per-mel-row min-max (the reference's behaviour)
    return D.log_on_mel(inputs)


@torch.no_grad()
def predict_frames(model, features: torch.Tensor, config, overlap_hop: int = 512, batch_size: int = 32,
                   smoothing: bool = True, threshold: bool = True) -> torch.Tensor:
    """features [M, T, C'] -> thresholded frame predictions [T, K] (metrics.py:56-81); `threshold=False` returns the
    averaged (and smoothed) probabilities the reference thresholds at 0.5 (metrics.py:81) instead.
    (For repeated evaluation pass `sj_train.InferenceEngine(model)` instead of the model: the same function with the
    eval-mode BatchNorms folded away and the HIP epilogues / LSTM launch, 1.4x the module's rate; or
    `sj_train.fold_batchnorm(model)` for the folding alone.)"""
    frame_len = features.shape[-2]
    windows = frame(features, config.n_frame, overlap_hop, pad_end=True, axis=-2)  # [M, W, n_frame, C']
    windows = windows.permute(1, 0, 2, 3)[..., :config.n_chan].contiguous()
    if hasattr(model, 'predict'):  # CustomModel: Keras-style predict (on a GPU: through its cached InferenceEngine)
        preds = model.predict(windows, batch_size=batch_size)
    else:
        model.eval()
        preds = torch.cat([model(windows[i:i + batch_size]) for i in range(0, windows.shape[0], batch_size)])
    if config.v in label_downsample_model:
        preds = preds.repeat_interleave(config.n_frame // preds.shape[-2], dim=-2)  # UpSampling1D
    preds = preds.permute(2, 0, 1)  # [K, W, n_frame]
    counts = overlap_and_add(torch.ones_like(preds), overlap_hop)[..., :frame_len]
    preds = overlap_and_add(preds, overlap_hop)[..., :frame_len] / counts
    preds = preds.t()
    if smoothing:
        preds = smooth(preds)
    return (preds >= 0.5).to(torch.float32) if threshold else preds


def evaluate_wav(model, wav, config, sample_rate: int = 16000, overlap_hop: int = 512, device=None) -> torch.Tensor:
    """In-memory counterpart of one iteration of metrics.evaluate's loop: [chan, samples]
    -> frame predictions [T, K]."""
    spec = D.load_wav_array(wav, sample_rate, device)
    return predict_frames(model, features_for_eval(spec, config), config, overlap_hop)
