"""Drop-in counterpart of the reference's pipeline.py (pipeline.py:6-175): synthesise a
training sample by mixing a background, up to N voices and noises in the complex-STFT
domain, and assemble the dataset graph.

Every random decision is drawn first (`merge_draw`, NumPy Generator, same
distributions as the reference's tf.random calls) and then applied deterministically
(`merge_complex_specs_apply`, torch ops on whatever device the tensors live on), so the
apply half can be checked against the oracle."""
from __future__ import annotations

from functools import partial
from typing import Optional

import numpy as np
import torch

from . import transforms as _tr
from .dataset import Dataset
from .utils import list_to_generator


def merge_draw(bg_frame: int, voice_frames, noise_frames=None, n_frame: int = 300, min_ratio: float = 2 / 3,
               min_noise_ratio: float = 1 / 2, snr: float = -20, rng: Optional[np.random.Generator] = None) -> dict:
    """All random draws of merge_complex_specs (pipeline.py:29-106).

    voice_frames / noise_frames: time lengths of the (padded) voices / noises.
      bg_offset ~ U{0 .. tiled_bg - n_frame}                    (tf.image.random_crop, :35)
      n_voices  ~ U{1 .. max_voices-1} (1 when max_voices == 1)  (:42-46)
      v_gain    = 10 ** -U[0, -snr/10)                           (:50)
      v_offset  ~ U{0 .. padded_len - n_frame - 1}               (:68-69; 0 when that range is empty,
                                                                  where TF would raise)
      n_noises  ~ U{0 .. max_noises-1}                           (:87-88)
      n_gain    = 10 ** -U[0, 2)                                 (:94)
      n_offset  ~ U{0 .. padded_len - n_frame}                   (tf.image.random_crop, :103)
    """
    rng = _tr.get_rng() if rng is None else rng
    reps = (n_frame + bg_frame - 1) // bg_frame
    d = {"bg_offset": int(rng.integers(0, reps * bg_frame - n_frame + 1))}
    max_voices = len(voice_frames)
    d["n_voices"] = int(rng.integers(1, max_voices)) if max_voices > 1 else 1
    d["v_gain"], d["v_offset"] = [], []
    for v in range(d["n_voices"]):
        d["v_gain"].append(float(np.float32(10.0) ** np.float32(-rng.uniform(0, -snr / 10))))
        vf = int(voice_frames[v])
        pad = n_frame - int(np.float32(min_ratio) * np.float32(vf))
        length = vf + 2 * pad if pad > 0 else vf
        maxval = length - n_frame
        d["v_offset"].append(int(rng.integers(0, maxval)) if maxval > 0 else 0)
    d["n_noises"], d["n_gain"], d["n_offset"] = 0, [], []
    if noise_frames is not None:
        d["n_noises"] = int(rng.integers(0, len(noise_frames))) if len(noise_frames) > 0 else 0
        for n in range(d["n_noises"]):
            d["n_gain"].append(float(np.float32(10.0) ** np.float32(-rng.uniform(0, 2))))
            nf = int(noise_frames[n])
            pad = n_frame - int(np.float32(min_noise_ratio) * np.float32(nf))
            length = nf + 2 * pad if pad > 0 else nf
            d["n_offset"].append(int(rng.integers(0, max(length - n_frame, 0) + 1)))
    return d


def merge_complex_specs_apply(background, voices, labels, noises, draws, n_frame=300, n_classes=3,
                              min_ratio=2 / 3, min_noise_ratio=1 / 2, seperate_noise_voice=False):
    """Deterministic part of merge_complex_specs (t_axis = 1).  Device tensors take the HIP path
    (`iris_mix_specs`, a handful of launches instead of ~15 torch ops per voice; with
    `seperate_noise_voice` the same kernels run over three source tables: everything, the voices alone,
    background + noises); CPU tensors use the op-by-op torch form below.  Both equal the oracle bit for bit."""
    if background.is_cuda and voices.is_cuda and (noises is None or noises.is_cuda):
        return _merge_apply_hip(background, voices, labels, noises, draws, n_frame, n_classes, min_ratio,
                                min_noise_ratio, seperate_noise_voice)
    bg_frame = background.shape[1]
    reps = (n_frame + bg_frame - 1) // bg_frame
    tiled = background.repeat(1, reps, 1)
    o = draws["bg_offset"]
    complex_spec = tiled[:, o:o + n_frame].clone()
    only_voice = torch.zeros_like(complex_spec)
    only_noise = complex_spec.clone()
    max_voices = voices.shape[0]
    label = torch.zeros((max_voices, n_frame, n_classes), dtype=torch.float32, device=complex_spec.device)
    for v in range(draws["n_voices"]):
        voice = voices[v]
        v_frame = voice.shape[1]
        l = labels[v:v + 1].to(torch.float32).repeat(v_frame, 1)
        active = (torch.amax(voice, dim=(0, 2)) > 0).to(torch.float32)
        l = l * active[:, None]
        pad = n_frame - int(np.float32(min_ratio) * np.float32(v_frame))
        if pad > 0:
            voice = torch.nn.functional.pad(voice, (0, 0, pad, pad))
            l = torch.nn.functional.pad(l, (0, 0, pad, pad))
        off = draws["v_offset"][v]
        voice = voice[:, off:off + n_frame]
        l = l[off:off + n_frame]
        l3 = torch.zeros_like(label)
        l3[v] = l
        no_overlap = (torch.amax(torch.sum(label + l3, dim=0)) < 2).to(torch.float32)
        gain = float(draws["v_gain"][v])
        complex_spec = complex_spec + gain * voice * no_overlap
        if seperate_noise_voice:
            only_voice = only_voice + gain * voice * no_overlap
        label = label + l3 * no_overlap
    if noises is not None:
        for n in range(draws["n_noises"]):
            noise = noises[n]
            ns_frame = noise.shape[1]
            pad = n_frame - int(np.float32(min_noise_ratio) * np.float32(ns_frame))
            if pad > 0:
                noise = torch.nn.functional.pad(noise, (0, 0, pad, pad))
            off = draws["n_offset"][n]
            noise = noise[:, off:off + n_frame]
            gain = float(draws["n_gain"][n])
            if seperate_noise_voice:
                only_noise = only_noise + gain * noise
            complex_spec = complex_spec + gain * noise
    if seperate_noise_voice:
        label = (label, only_voice, only_noise)
    return complex_spec, label


def _merge_apply_hip(background, voices, labels, noises, draws, n_frame, n_classes, min_ratio, min_noise_ratio,
                     seperate_noise_voice=False):
    """One sample through the batched synthesis kernels (include/iris_frontend.h: iris_mix_specs).
    `seperate_noise_voice` (pipeline.py:38-39, :80-81, :104-108): only_voice is the sum over a table holding the
    voices alone (accepted by the same label rule; 0 + x is exact), only_noise the sum over background + noises -
    each the same op-by-op order as the reference's running sums."""
    import ctypes as C

    from . import _native as N
    from .mixer import KIND_BACKGROUND, KIND_NOISE, KIND_VOICE, MIX_SRC
    dev = background.device
    background = background.to(torch.float32).contiguous()
    voices = voices.to(torch.float32).contiguous()
    label_vecs = labels.to(device=dev, dtype=torch.float32).contiguous()
    n_bins, chan2 = int(background.shape[0]), int(background.shape[2])
    max_voices, v_frame = int(voices.shape[0]), int(voices.shape[2])
    lib = N.lib()
    keep = [background, voices, label_vecs]
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        recs = [(background.data_ptr(), 0, int(background.shape[1]), 0, int(draws["bg_offset"]), 1.0,
                 KIND_BACKGROUND, 0, 0, 0)]
        pad = n_frame - int(np.float32(min_ratio) * np.float32(v_frame))
        active = torch.empty((max(draws["n_voices"], 1), v_frame), device=dev, dtype=torch.float32)
        keep.append(active)
        for v in range(draws["n_voices"]):
            N.check(lib.iris_mix_frame_active(voices[v].data_ptr(), n_bins, v_frame, chan2, active[v].data_ptr(), stream),
                    "iris_mix_frame_active")
            recs.append((voices[v].data_ptr(), active[v].data_ptr(), v_frame, max(pad, 0), int(draws["v_offset"][v]),
                         np.float32(draws["v_gain"][v]), KIND_VOICE, v, v, 0))
        if noises is not None and draws["n_noises"]:
            noises = noises.to(torch.float32).contiguous()
            keep.append(noises)
            ns_frame = int(noises.shape[2])
            pad = n_frame - int(np.float32(min_noise_ratio) * np.float32(ns_frame))
            for n in range(draws["n_noises"]):
                recs.append((noises[n].data_ptr(), 0, ns_frame, max(pad, 0), int(draws["n_offset"][n]),
                             np.float32(draws["n_gain"][n]), KIND_NOISE, 0, 0, 0))

        def mix(rows):
            table = np.array(rows, dtype=MIX_SRC)
            table_d = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(dev)
            first_d = torch.tensor([0, len(rows)], dtype=torch.int32, device=dev)
            spec = torch.empty((1, n_bins, n_frame, chan2), device=dev, dtype=torch.float32)
            label = torch.empty((1, max_voices, n_frame, n_classes), device=dev, dtype=torch.float32)
            ws_floats = int(lib.iris_mix_workspace(len(rows), n_frame))
            ws = torch.empty(max(ws_floats, 1), device=dev, dtype=torch.float32)
            N.check(lib.iris_mix_specs(table_d.data_ptr(), len(rows), first_d.data_ptr(), label_vecs.data_ptr(),
                                       spec.data_ptr(), label.data_ptr(), 1, n_bins, n_frame, chan2, max_voices,
                                       n_classes, ws.data_ptr(), ws_floats, stream), "iris_mix_specs")
            keep.extend([table_d, first_d, ws])
            return spec[0], label[0]

        spec, label = mix(recs)
        if seperate_noise_voice:
            only_voice, _ = mix([r for r in recs if r[6] == KIND_VOICE])
            only_noise, _ = mix([r for r in recs if r[6] != KIND_VOICE])
            label = (label, only_voice, only_noise)
        for t in keep:
            t.record_stream(torch.cuda.current_stream(dev))
    return spec, label


def merge_complex_specs(background, voices_and_labels, noises=None, n_frame=300, n_classes=3, t_axis=1,
                        min_ratio=2 / 3, min_noise_ratio=1 / 2, snr=-20, seperate_noise_voice=False):
    """OUTPUT: complex_spec (freq, time, chan2), labels (n_voices, time, n_classes)
    (pipeline.py:6-110)."""
    if t_axis != 1:
        raise ValueError("only t_axis=1 ([freq, time, chan2]) is supported")
    voices, labels = voices_and_labels
    draws = merge_draw(background.shape[1], [voices.shape[2]] * voices.shape[0],
                       None if noises is None else [noises.shape[2]] * noises.shape[0],
                       n_frame, min_ratio, min_noise_ratio, snr)
    return merge_complex_specs_apply(background, voices, labels, noises, draws, n_frame, n_classes, min_ratio,
                                     min_noise_ratio, seperate_noise_voice)


def make_pipeline(backgrounds, voices, labels, noises=None, n_frame=300, max_voices=10, max_noises=10,
                  n_classes=3, device=None, **kwargs):
    """Dataset of (complex spectrogram [freq_bins, n_frame, chan*2], labels
    [max_voices, n_frame, n_classes]) (pipeline.py:113-175).  `device`: where the mixing
    runs (default: the current ROCm device when there is one)."""
    assert len(backgrounds[0].shape) == 3, 'each spec must be a 3D-tensor'
    assert len(voices) == len(labels)
    assert len(labels[0].shape) == 1 and labels[0].shape[0] == n_classes, \
        'labels must be in the form of [n_samples, n_classes]'
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)

    def to_dev(x):
        return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(device, non_blocking=True)

    def resident(data):
        """Every source goes to the device once, here (the corpora fit in HBM many times over), not once
        per sample it is drawn for."""
        if isinstance(data, tuple):
            return tuple(resident(d) for d in data)
        return [to_dev(x) for x in data]

    def gen_of(data):
        return list_to_generator(resident(data))

    b_dataset = Dataset.from_generator(gen_of(backgrounds)).repeat().shuffle(len(backgrounds))
    v_dataset = Dataset.from_generator(gen_of((voices, labels))).repeat().shuffle(len(voices))
    v_dataset = v_dataset.padded_batch(max_voices)
    if noises is not None:
        n_dataset = Dataset.from_generator(gen_of(noises)).repeat().shuffle(len(noises))
        n_dataset = n_dataset.padded_batch(max_noises)
        dataset = Dataset.zip((b_dataset, v_dataset, n_dataset))
    else:
        dataset = Dataset.zip((b_dataset, v_dataset))
    return dataset.map(partial(merge_complex_specs, n_frame=n_frame, n_classes=n_classes, **kwargs))
