"""BASELINE configs[4]: 22.05 kHz stereo, n_fft 2048, hop 512, 128 mel - the banded fp32 mel kernel and the
fp16-MFMA variant side by side (event-timed dominant kernel + step time).
usage: python scripts/gpu_c5.py [fp32|fp16_mfma|both] [launches]      (under rocprofv3 --pmc: one variant per run)"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from challenge_amd.frontend import FrontendPlan, normalize

SR, N_FFT, HOP, M, C, B, SECONDS = 22050, 2048, 512, 128, 2, 16, 10


def run(variant, n):
    dev = torch.device("cuda", 0)
    length = SR * SECONDS
    plan = FrontendPlan(N_FFT, HOP, M, SR, C, B, length, dev)
    plan.set_mel_precision(variant)
    gen = torch.Generator(device=dev).manual_seed(5)
    copies = 6  # 6 x (35.3 MB in + 14.1 MB out) = 296 MB per cycle > 256 MiB Infinity Cache
    wavs = [normalize(torch.randn(B, C, length, generator=gen, device=dev)) for _ in range(copies)]
    outs = [torch.empty((B, M, plan.num_frames(length), C), device=dev) for _ in range(copies)]
    for i in range(copies):
        plan.wav_to_logmel(wavs[i], out=outs[i])
    plan.timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        plan.wav_to_logmel(wavs[i % copies], out=outs[i % copies])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    n_ev, k_ms = plan.timing_read()
    frames = B * C * plan.num_frames(length)
    algo = B * (C * length * 4 + M * plan.num_frames(length) * C * 4)     # 220,500 B per stereo audio-second
    res = {"variant": variant, "k1_us": round(1e3 * k_ms, 2), "step_us_with_event_pairs": round(1e6 * dt, 2),
           "audio_s_per_s": round(B * SECONDS / dt, 1), "frame_channels": frames,
           "k1_frac_of_8TBs": round(algo / (k_ms * 1e-3) / 8e12, 4)}
    if variant == "fp16_mfma":
        # dense-equivalent and issued flops of the contraction: 2 F M per frame-channel dense;
        # issued = MFMAs x 16x16x32 x 2 (block-sparse: only k-steps holding a non-zero of a 16-band tile)
        w = plan.mel_matrix
        n_mfma_per_group = 0
        for t in range((M + 15) // 16):
            nz = (w[:, 16 * t:16 * t + 16] != 0).any(axis=1).nonzero()[0]
            if len(nz):
                n_mfma_per_group += (nz.max() // 32) - (nz.min() // 32) + 1
        groups = frames / 8
        issued = n_mfma_per_group * groups * 2 * 16 * 16 * 32
        res.update(mfma_per_group_of_8_frames=int(n_mfma_per_group), issued_tflops=round(issued / (k_ms * 1e-3) / 1e12, 3),
                   dense_equiv_tflops=round(frames * 2 * 1025 * M / (k_ms * 1e-3) / 1e12, 3),
                   mfma_pipe_util_of_2500TF=round(issued / (k_ms * 1e-3) / 2.5e15, 5))
    return res


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    for v in (["fp32", "fp16_mfma"] if which == "both" else [which]):
        print(json.dumps(run(v, n)), flush=True)
