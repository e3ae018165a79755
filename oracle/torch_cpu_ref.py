"""CPU baseline 'B' -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

The engine the reference really executes for the STFT is torch.stft (through
torchaudio.transforms.Spectrogram, data_utils.py:17-23); the rest of the chain
(transforms.py:111-123, :58-70; data_utils.py:37-55) is restated with torch CPU
ops so the whole path runs multi-threaded on the host cores.  Used only by
bench.py's cpu_baseline leg and by tests (cross-checked against the NumPy oracle).
"""
import torch


def wav_to_logmel_cpu(wav: torch.Tensor, w: torch.Tensor, n_fft: int, hop: int,
                      do_minmax: bool = True, do_log: bool = True) -> torch.Tensor:
    """wav [B,C,L] fp32 CPU, w [F,M] -> [B,M,T,C]."""
    b, c, length = wav.shape
    spec = torch.stft(wav.reshape(b * c, length), n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.hann_window(n_fft), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)  # [BC,F,T]
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2).reshape(b, c, spec.shape[1], spec.shape[2])
    mel = torch.einsum("bcft,fm->bmtc", mag, w)
    if do_minmax:
        flat = mel.reshape(b, -1)
        mn = flat.min(dim=1).values.reshape(b, 1, 1, 1)
        mx = flat.max(dim=1).values.reshape(b, 1, 1, 1)
        mel = (mel - mn) / torch.clamp(mx - mn, min=1e-8)
    if do_log:
        mel = torch.log(mel + 1e-8)
    return mel
