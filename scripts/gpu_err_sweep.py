"""Seeded accuracy sweep of the fused HIP kernel's mel (before min-max / log) against the fp64 oracle
("next" item 1): >= 50 seeds x {reference default shape at full size, n_fft 256, c2, c3, c5},
amplitude 0.01 .. 2 (log-uniform per seed), with and without SpecAugment / stft_filter bands.  Beside the HIP kernel,
on the first --engines seeds, three CPU engines on the same input: scipy's fp32 pocketfft (a genuine fp32 FFT),
torch.stft fp32 (what the reference runs through torchaudio) and NumPy's float32 rfft (numpy 2.x: fp64 inside, rounded
once - listed to show that it is no fp32 yardstick).  Two metrics per case:

    old   max |d| / max(|ref|, 1e-3)                    SURVEY section 8(d)'s metric
    rule  max |d| / (1e-5 |ref| + 4 eps xrms[b,t,c] sum_k W[k,m])   oracle.frontend_ref.mel_tolerance (<= 1 passes)

and `noise`: max |d| / (eps xrms wsum) over elements with |ref| < 1e-3 (the empirical constant of the rule's second
term).  Output: one line per (shape, seed, bands), then per-shape worst cases.

    python scripts/gpu_err_sweep.py [--seeds 50] [--engines 10] > profiles/r4/hip_vs_fp64_sweep.log
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.fft as sfft
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import frontend_ref as R  # noqa: E402
from oracle.torch_cpu_ref import wav_to_logmel_cpu  # noqa: E402
from challenge_amd.frontend import FrontendPlan  # noqa: E402

# name, n_fft, hop, n_mel, channels, sample rate, samples, batch (first seeds), batch (remaining seeds), bands (time n x max, freq n x max, filter)
SHAPES = [
    ("refdefault", 512, 256, 80, 2, 16000, 511 * 256, 12, 12, (6, 23, 1, 15, 3)),   # sj_train.py:46,59; data_utils.py:17,58-61
    ("n256", 256, 128, 40, 1, 16000, 16000, 8, 8, (3, 12, 1, 8, 2)),
    ("c2", 1024, 256, 64, 1, 16000, 160000, 32, 8, (6, 23, 1, 15, 3)),
    ("c3", 1024, 256, 64, 1, 16000, 130816, 64, 8, (6, 23, 1, 15, 3)),
    ("c5", 2048, 512, 128, 2, 22050, 220500, 4, 2, (6, 23, 1, 30, 3)),
]


def cpu_mels(wav, n_fft, hop, m, sr, t_bands, f_bands):
    """mel of the three CPU engines, bands applied to the magnitudes like the oracle does."""
    w = R.linear_to_mel_weight_matrix(m, n_fft // 2 + 1, sr)
    fr = (R.frame_signal(wav, n_fft, hop) * R.hann_periodic(n_fft)).astype(np.float32)  # [B,C,T,N]
    mags = {"scipy_fp32": np.abs(sfft.rfft(fr, axis=-1)).astype(np.float32),
            "numpy_f32(fp64 inside)": np.abs(np.fft.rfft(fr, axis=-1)).astype(np.float32)}
    b, c, length = wav.shape
    spec = torch.stft(torch.from_numpy(wav).reshape(b * c, length), n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.hann_window(n_fft), center=True, pad_mode="reflect", return_complex=True)
    mags["torch_stft_fp32"] = spec.abs().numpy().reshape(b, c, n_fft // 2 + 1, -1).transpose(0, 1, 3, 2).copy()
    out = {}
    for name, mag in mags.items():
        if t_bands is not None:
            for i in range(b):
                for off, size in t_bands[i]:
                    mag[i, :, off:off + size, :] = 0
        if f_bands is not None:
            for i in range(b):
                for off, size in f_bands[i]:
                    mag[i, :, :, off:off + size] = 0
        out[name] = np.einsum("bctf,fm->bmtc", mag, w, optimize=True)
    return out


def metrics(mel, ref, tol, noise_unit):
    d = np.abs(mel.astype(np.float64) - ref)
    old = float((d / np.maximum(np.abs(ref), 1e-3)).max())
    rule = R.mel_err_ratio(mel, ref, tol)
    small = (np.abs(ref) < 1e-3) & (noise_unit > 0)
    noise = float((d[small] / noise_unit[small]).max()) if small.any() else 0.0
    return old, rule, noise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=50)
    ap.add_argument("--engines", type=int, default=10)
    ap.add_argument("--full", type=int, default=5, help="seeds run at the configuration's full batch")
    ap.add_argument("--shapes", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    eps = float(np.finfo(np.float32).eps)
    worst = {}
    t_start = time.time()
    for name, n_fft, hop, m, c, sr, length, b_full, b_rest, (ntb, tmax, nfb, fmax, filt) in SHAPES:
        if a.shapes and name not in a.shapes.split(","):
            continue
        n_t, n_f = 1 + length // hop, n_fft // 2 + 1
        plans = {}
        for seed in range(a.seeds):
            b = b_full if seed < a.full else b_rest
            if b not in plans:
                plans[b] = FrontendPlan(n_fft, hop, m, sr, c, b, length, dev)
            rng = np.random.default_rng(10_000 * (1 + SHAPES.index(next(s for s in SHAPES if s[0] == name))) + seed)
            amp = float(np.exp(rng.uniform(np.log(0.01), np.log(2.0))))
            wav = (rng.standard_normal((b, c, length)) * amp).astype(np.float32)
            # SpecAugment draws of `augment` (data_utils.py:58-61) + stft_filter(k) (bins 1..k, data_utils.py:126-136)
            tb = np.stack([np.stack(R.mask_draw(rng, n_t, tmax, ntb), 1) for _ in range(b)])
            fb = np.stack([np.concatenate([np.stack(R.mask_draw(rng, n_f, fmax, nfb), 1), np.array([[1, filt]], np.int32)])
                           for _ in range(b)])
            x = torch.from_numpy(wav).to(dev)
            for tag, kw in (("plain", {}), ("bands", {"t_bands": tb, "f_bands": fb})):
                ref, tol = R.mel_tolerance(wav, n_fft, hop, m, sr, **kw)
                noise_unit = tol - R.MEL_REL_TOL * np.abs(ref)  # = ulps eps xrms wsum
                noise_unit = noise_unit / R.MEL_NOISE_ULPS
                hip = plans[b].wav_to_logmel(x, minmax=False, log=False, **kw).cpu().numpy()
                res = {"hip": metrics(hip, ref, tol, noise_unit)}
                if seed < a.engines:
                    for en, mel in cpu_mels(wav, n_fft, hop, m, sr, kw.get("t_bands"), kw.get("f_bands")).items():
                        res[en] = metrics(mel, ref, tol, noise_unit)
                line = f"{name:10s} seed {seed:2d} B {b:2d} amp {amp:6.3f} {tag:5s} mel [{ref.min():.3g}, {ref.max():.3g}]"
                for en, (old, rule, noise) in res.items():
                    line += f" | {en}: old {old:.2e} rule {rule:.3f} noise {noise:.2f}eps"
                    w_ = worst.setdefault((name, en), [0.0, 0.0, 0.0])
                    w_[0], w_[1], w_[2] = max(w_[0], old), max(w_[1], rule), max(w_[2], noise)
                print(line, flush=True)
        del plans
    print(f"\n# worst over the sweep ({a.seeds} seeds x plain / bands; CPU engines on the first {a.engines} seeds); "
          f"rule = |d| <= {R.MEL_REL_TOL:g} |ref| + {R.MEL_NOISE_ULPS:g} eps xrms wsum; {time.time() - t_start:.0f} s")
    for (name, en), (old, rule, noise) in worst.items():
        print(f"{name:10s} {en:24s} old-metric {old:.2e}   rule {rule:.3f}   noise-term constant {noise:.2f} eps")


if __name__ == "__main__":
    main()
