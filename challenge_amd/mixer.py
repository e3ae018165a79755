"""GPU-resident sample synthesis: make_pipeline + merge_complex_specs for whole batches.

The reference builds every training sample on the host, one at a time: a tf.data graph
picks a background, up to N voices and noises and mixes them in the complex-STFT domain
(pipeline.py:6-175).  On an MI355X every source spectrogram of the corpus fits in HBM, so
this module keeps them resident and synthesises a whole batch with two kernel launches
(`iris_mix_specs`, include/iris_frontend.h): the host only draws the random decisions
(`pipeline.merge_draw`, same distributions as the reference) and uploads a table of a few
dozen bytes per source.

No CPU fallback: `DeviceMixer` needs a ROCm device and the HIP library.  The per-sample
drop-in (`pipeline.make_pipeline`) stays available for code that wants the tf.data shape.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N
from . import pipeline as _pl

# mirrors iris_mix_src (include/iris_frontend.h)
MIX_SRC = np.dtype([("src", "<u8"), ("active", "<u8"), ("T", "<i4"), ("pad", "<i4"), ("off", "<i4"),
                    ("gain", "<f4"), ("kind", "<i4"), ("slot", "<i4"), ("label_row", "<i4"), ("reserved", "<i4")])
assert MIX_SRC.itemsize == 48

KIND_BACKGROUND, KIND_VOICE, KIND_NOISE = 0, 1, 2


class _Stream:
    """`Dataset.from_generator(data).repeat().shuffle(len(data))` as an index stream: a fresh
    random permutation per epoch (pipeline.py:147-160)."""

    def __init__(self, n: int, rng: np.random.Generator):
        self.n, self.rng = n, rng
        self._perm = np.empty(0, np.int64)
        self._pos = 0

    def take(self, k: int) -> List[int]:
        out = []
        while len(out) < k:
            if self._pos >= len(self._perm):
                self._perm, self._pos = self.rng.permutation(self.n), 0
            out.append(int(self._perm[self._pos]))
            self._pos += 1
        return out


class DeviceMixer:
    """Batched, device-resident counterpart of `make_pipeline(...)` (pipeline.py:113-175):

        mixer = DeviceMixer(backgrounds, voices, labels, noises, n_frame=512, ...)
        spec, label = mixer.mix(batch)     # [B, F, n_frame, 2C], [B, max_voices, n_frame, n_classes]

    backgrounds / voices / noises: sequences of [F, T_i, 2C] arrays (ragged in T); labels:
    [n_voices, n_classes] rows (one-hot in the reference).  Source picking follows the
    reference's dataset graph: one background, the next `max_voices` voices and the next
    `max_noises` noises of shuffled, repeated streams; each group is zero-padded to its longest
    member (`padded_batch`), of which `merge_complex_specs` uses the first n_voices / n_noises.
    """

    def __init__(self, backgrounds: Sequence, voices: Sequence, labels, noises: Optional[Sequence] = None,
                 n_frame: int = 300, max_voices: int = 10, max_noises: int = 10, n_classes: int = 3, device=None,
                 min_ratio: float = 2 / 3, min_noise_ratio: float = 1 / 2, snr: float = -20, seed=None):
        labels = np.asarray(labels, np.float32)
        assert len(np.asarray(backgrounds[0]).shape) == 3, 'each spec must be a 3D-tensor'
        assert len(voices) == len(labels)
        assert labels.ndim == 2 and labels.shape[1] == n_classes, \
            'labels must be in the form of [n_samples, n_classes]'
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("DeviceMixer needs a ROCm device (no CPU fallback); use pipeline.make_pipeline")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceMixer needs a ROCm device (no CPU fallback); use pipeline.make_pipeline")
        N.lib()  # fail loudly when the HIP library is missing
        self.n_frame, self.max_voices, self.max_noises, self.n_classes = n_frame, max_voices, max_noises, n_classes
        self.min_ratio, self.min_noise_ratio, self.snr = min_ratio, min_noise_ratio, snr
        self.rng = np.random.default_rng(seed)

        def upload(items):
            out = [torch.as_tensor(np.ascontiguousarray(np.asarray(x, np.float32))).to(self.device) for x in items]
            for t in out:
                if t.dim() != 3 or t.shape[0] != out[0].shape[0] or t.shape[2] != out[0].shape[2]:
                    raise ValueError("sources must be [freq, time, chan2] with equal freq and chan2")
            return out

        self.backgrounds, self.voices = upload(backgrounds), upload(voices)
        self.noises = upload(noises) if noises is not None else None
        self.n_bins, self.chan2 = int(self.backgrounds[0].shape[0]), int(self.backgrounds[0].shape[2])
        for group in (self.voices, self.noises or []):
            for t in group:
                if t.shape[0] != self.n_bins or t.shape[2] != self.chan2:
                    raise ValueError("voices / noises must share the backgrounds' freq and chan2 sizes")
        self.label_vecs = torch.from_numpy(labels).to(self.device)
        # which frames of a voice are active (max over freq, chan2 > 0; pipeline.py:57) is a property of
        # the source: one pass over the corpus now instead of one per use
        self.voice_active = []
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            for v in self.voices:
                act = torch.empty(int(v.shape[1]), device=self.device, dtype=torch.float32)
                N.check(N.lib().iris_mix_frame_active(v.data_ptr(), self.n_bins, int(v.shape[1]), self.chan2,
                                                      act.data_ptr(), stream), "iris_mix_frame_active")
                self.voice_active.append(act)
        # (pointer, frames) of every source, looked up per record when a batch's table is built
        self._bg_rec = [(t.data_ptr(), int(t.shape[1])) for t in self.backgrounds]
        self._v_rec = [(t.data_ptr(), a.data_ptr(), int(t.shape[1])) for t, a in zip(self.voices, self.voice_active)]
        self._n_rec = [(t.data_ptr(), int(t.shape[1])) for t in self.noises] if self.noises is not None else []
        self._b = _Stream(len(self.backgrounds), self.rng)
        self._v = _Stream(len(self.voices), self.rng)
        self._n = _Stream(len(self.noises), self.rng) if self.noises is not None else None

    # -- random half ------------------------------------------------------------------
    def draw(self, batch: int) -> List[dict]:
        """Per sample: which sources (dataset graph, pipeline.py:147-174) and the draws of
        merge_complex_specs (`pipeline.merge_draw`, pipeline.py:29-106)."""
        out = []
        for _ in range(batch):
            bg = self._b.take(1)[0]
            vs = self._v.take(self.max_voices)
            ns = self._n.take(self.max_noises) if self._n is not None else None
            v_len = max(self._v_rec[i][2] for i in vs)       # padded_batch: longest of the group
            n_len = max(self._n_rec[i][1] for i in ns) if ns else 0
            d = _pl.merge_draw(self._bg_rec[bg][1], [v_len] * len(vs),
                               [n_len] * len(ns) if ns is not None else None, self.n_frame, self.min_ratio,
                               self.min_noise_ratio, self.snr, rng=self.rng)
            d.update(bg=bg, voices=vs, noises=ns, v_len=v_len, n_len=n_len)
            out.append(d)
        return out

    # -- deterministic half ----------------------------------------------------------
    def table(self, draws: List[dict]) -> Tuple[np.ndarray, np.ndarray]:
        """Source table (iris_mix_src records) and the per-sample ranges for a list of draws."""
        recs, first = [], [0]
        for d in draws:
            ptr, frames = self._bg_rec[d["bg"]]
            recs.append((ptr, 0, frames, 0, int(d["bg_offset"]), 1.0, KIND_BACKGROUND, 0, 0, 0))
            pad = self.n_frame - int(np.float32(self.min_ratio) * np.float32(d["v_len"]))
            for v in range(d["n_voices"]):
                ptr, act, frames = self._v_rec[d["voices"][v]]
                recs.append((ptr, act, frames, max(pad, 0), int(d["v_offset"][v]), np.float32(d["v_gain"][v]),
                             KIND_VOICE, v, d["voices"][v], 0))
            if d["noises"] is not None:
                pad = self.n_frame - int(np.float32(self.min_noise_ratio) * np.float32(d["n_len"]))
                for n in range(d["n_noises"]):
                    ptr, frames = self._n_rec[d["noises"][n]]
                    recs.append((ptr, 0, frames, max(pad, 0), int(d["n_offset"][n]), np.float32(d["n_gain"][n]),
                                 KIND_NOISE, 0, 0, 0))
            first.append(len(recs))
        return np.array(recs, dtype=MIX_SRC), np.asarray(first, np.int32)

    def mix(self, batch: int, draws: Optional[List[dict]] = None):
        """One batch of (complex spectrogram [B, F, n_frame, 2C], labels [B, max_voices, n_frame,
        n_classes]) - `merge_complex_specs` (pipeline.py:6-110) for every sample, two launches."""
        draws = self.draw(batch) if draws is None else draws
        batch = len(draws)
        table, first = self.table(draws)
        n_srcs = int(table.shape[0])
        dev = self.device
        table_d = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(dev, non_blocking=True)
        first_d = torch.from_numpy(first).to(dev, non_blocking=True)
        spec = torch.empty((batch, self.n_bins, self.n_frame, self.chan2), device=dev, dtype=torch.float32)
        label = torch.empty((batch, self.max_voices, self.n_frame, self.n_classes), device=dev, dtype=torch.float32)
        ws_floats = int(N.lib().iris_mix_workspace(n_srcs, self.n_frame))
        ws = torch.empty(max(ws_floats, 1), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            rc = N.lib().iris_mix_specs(table_d.data_ptr(), n_srcs, first_d.data_ptr(), self.label_vecs.data_ptr(),
                                        spec.data_ptr(), label.data_ptr(), batch, self.n_bins, self.n_frame,
                                        self.chan2, self.max_voices, self.n_classes, ws.data_ptr(), ws_floats,
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        N.check(rc, "iris_mix_specs")
        # the table, ranges and workspace must outlive the kernels: tie them to the stream
        for t in (table_d, first_d, ws):
            t.record_stream(torch.cuda.current_stream(dev))
        return spec, label

    def __iter__(self):
        raise TypeError("DeviceMixer yields whole batches: call mix(batch)")
