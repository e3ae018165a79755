"""Generate the committed golden fixtures for the frontend parity tests.

Run once in the build container (torch 2.10 CPU available, no TF, no torchaudio):

    python tests/golden/make_golden.py

What is pinned and by what
--------------------------
* STFT: ``torch.stft`` called exactly as ``torchaudio.transforms.Spectrogram(
  n_fft, power=None)`` calls it (reference data_utils.py:17,:23): win_length =
  n_fft, hop = n_fft//2 (or the config's hop), periodic Hann, center=True,
  pad_mode='reflect', normalized=False, onesided=True.  This is the engine the
  reference runs, executed here; its outputs are stored.
* magnitude -> mel -> min-max -> log: computed from the torch.stft spectrum with
  plain NumPy fp32 following transforms.py:51-77,:111-123 and
  data_utils.py:37-55.  The mel matrix is this repo's restatement of TF's recipe
  (parity unpinned, see oracle/frontend_ref.py header), stored alongside so the
  GPU tests and the oracle share one matrix.
The script imports nothing from /root/reference.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import frontend_ref as R  # noqa: E402  (for the mel matrix + normalize only)


def torch_spectrogram(wav, n_fft, hop):
    x = torch.from_numpy(wav)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.hann_window(n_fft), center=True,
                      pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    return spec.numpy()  # [C, F, T] complex64


def make_case(name, seed, chans, length, n_fft, hop, n_mel, sr, keep_frames):
    rng = np.random.default_rng(seed)
    wav = rng.standard_normal((chans, length)).astype(np.float32)
    wav = R.normalize(wav)
    spec = torch_spectrogram(wav, n_fft, hop)  # [C,F,T]
    f, t = spec.shape[1], spec.shape[2]
    mag = np.abs(spec).astype(np.float32)  # sqrt(re^2+im^2)
    w = R.linear_to_mel_weight_matrix(n_mel, f, sr)
    mel = np.einsum("cft,fm->mtc", mag, w).astype(np.float32)  # [M,T,C]
    mel_b = mel[None]
    mn = mel_b.min(axis=(1, 2, 3), keepdims=True)
    mx = mel_b.max(axis=(1, 2, 3), keepdims=True)
    norm = (mel_b - mn) / np.maximum(mx - mn, np.float32(1e-8))
    logmel = np.log(norm + np.float32(1e-8)).astype(np.float32)[0]
    frames = np.array(sorted(set(keep_frames(t))), dtype=np.int32)
    out = dict(wav=wav, n_fft=n_fft, hop=hop, n_mel=n_mel, sample_rate=sr,
               n_frames=t, spec_frames=frames,
               spec_re=spec.real[:, :, frames].astype(np.float32),
               spec_im=spec.imag[:, :, frames].astype(np.float32),
               mel=mel, logmel=logmel)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, "wav", wav.shape, "spec", spec.shape, "mel", mel.shape,
          "mel range", float(mel.min()), float(mel.max()))


def main():
    edge = lambda t: [0, 1, 2, 3, t // 2, t - 3, t - 2, t - 1]
    # c1 (BASELINE.json configs[0]): one 2 s mono 16 kHz clip, n_fft 1024 hop 256 mel 64
    make_case("c1_mono_2s", 1234, 1, 32000, 1024, 256, 64, 16000, edge)
    # reference default shape: n_fft 512 (hop 256), 80 mel, stereo (data_utils.py:17, sj_train.py:46)
    make_case("refdefault_stereo", 4321, 2, 12345, 512, 256, 80, 16000, edge)
    # c5 shape, short: 22.05 kHz stereo n_fft 2048 hop 512 mel 128
    make_case("c5_stereo_short", 777, 2, 11025, 2048, 512, 128, 22050, edge)
    # ragged length (L not a multiple of hop), n_fft 256
    make_case("ragged_n256", 99, 1, 3001, 256, 128, 40, 16000, edge)

    # mel matrices in sparse form, for the three BASELINE shapes
    mats = {}
    for (m, f, sr) in [(80, 257, 16000), (64, 513, 16000), (128, 1025, 22050), (40, 129, 16000)]:
        w = R.linear_to_mel_weight_matrix(m, f, sr)
        nz = np.nonzero(w)
        mats[f"w_{m}_{f}_{sr}_rows"] = nz[0].astype(np.int32)
        mats[f"w_{m}_{f}_{sr}_cols"] = nz[1].astype(np.int32)
        mats[f"w_{m}_{f}_{sr}_vals"] = w[nz]
        print("mel", m, f, sr, "nnz", len(nz[0]), "max nnz per bin", int((w > 0).sum(1).max()))
    np.savez(os.path.join(HERE, "mel_matrices.npz"), **mats)


if __name__ == "__main__":
    main()
