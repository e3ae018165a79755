"""HIP mel (before min-max / log) against the fp64 oracle, beside what an fp32 torch.stft pipeline (the engine the
reference runs) achieves on the same input: the honest reading of north_star's 1e-5."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import frontend_ref as R
from oracle.torch_cpu_ref import wav_to_logmel_cpu
from challenge_amd.frontend import FrontendPlan


def rel_err(a, ref, floor=1e-3):
    return float((np.abs(a - ref) / np.maximum(np.abs(ref), floor)).max())


dev = torch.device("cuda", 0)
rng = np.random.default_rng(77)
for n_fft, hop, m, c, b, length, sr in [(1024, 256, 64, 1, 40, 25600, 16000), (512, 256, 80, 2, 24, 20000, 16000),
                                        (2048, 512, 128, 2, 4, 44100, 22050), (256, 128, 40, 1, 8, 16000, 16000),
                                        (1024, 256, 64, 1, 4, 160000, 16000)]:
    wav = (rng.standard_normal((b, c, length)) * 0.1).astype(np.float32)
    ref64 = R.wav_to_mel(wav, n_fft, hop, m, sr, dtype=np.float64)
    ref32 = R.wav_to_mel(wav, n_fft, hop, m, sr)
    w = torch.from_numpy(R.linear_to_mel_weight_matrix(m, n_fft // 2 + 1, sr))
    t32 = wav_to_logmel_cpu(torch.from_numpy(wav), w, n_fft, hop, False, False).numpy()
    plan = FrontendPlan(n_fft, hop, m, sr, c, b, length, dev)
    hip = plan.wav_to_logmel(torch.from_numpy(wav).to(dev), minmax=False, log=False).cpu().numpy()
    print(f"n_fft {n_fft} M {m} C {c}: HIP vs fp64 {rel_err(hip, ref64):.2e} | numpy fp32 vs fp64 {rel_err(ref32, ref64):.2e} | "
          f"torch.stft fp32 vs fp64 {rel_err(t32, ref64):.2e} | HIP vs numpy fp32 {rel_err(hip, ref32):.2e} | mel range [{ref64.min():.3g}, {ref64.max():.3g}]")


# the two inputs of tests/test_frontend_gpu.py::test_workgroups_looping_over_several_chunks (same generator sequence)
rng = np.random.default_rng(77)
for n_fft, hop, m, c, b, length in [(1024, 256, 64, 1, 40, 25600), (512, 256, 80, 2, 24, 20000)]:
    wav = (rng.standard_normal((b, c, length)) * 0.1).astype(np.float32)
    n_t, n_f = 1 + length // hop, n_fft // 2 + 1
    tb = np.stack([np.stack(R.mask_draw(rng, n_t, 12, 3), 1) for _ in range(b)])
    fb = np.stack([np.stack(R.mask_draw(rng, n_f, 24, 2), 1) for _ in range(b)])
    plan = FrontendPlan(n_fft, hop, m, 16000, c, b, length, dev)
    w = torch.from_numpy(R.linear_to_mel_weight_matrix(m, n_f, 16000))
    t32 = wav_to_logmel_cpu(torch.from_numpy(wav), w, n_fft, hop, False, False).numpy()
    for name, kw in (("no bands", {}), ("t", {"t_bands": tb}), ("t+f", {"t_bands": tb, "f_bands": fb}), ("f", {"f_bands": fb})):
        ref64 = R.wav_to_mel(wav, n_fft, hop, m, 16000, dtype=np.float64, **kw)
        ref32 = R.wav_to_mel(wav, n_fft, hop, m, 16000, **kw)
        hip = plan.wav_to_logmel(torch.from_numpy(wav).to(dev), minmax=False, log=False, **kw).cpu().numpy()
        extra = f" | torch.stft fp32 vs fp64 {rel_err(t32, ref64):.2e}" if not kw else ""
        print(f"test input n_fft {n_fft} M {m} C {c} [{name}]: HIP vs fp64 {rel_err(hip, ref64):.2e} | numpy fp32 vs fp64 "
              f"{rel_err(ref32, ref64):.2e}{extra}")

# the three inputs of tests/test_frontend_gpu.py::test_fp16_mfma_mel_variant (fp32 path)
for n_fft, hop, m, c, sr, b, length in [(2048, 512, 128, 2, 22050, 3, 33075), (1024, 256, 64, 1, 16000, 5, 40000), (512, 256, 80, 2, 16000, 4, 20000)]:
    rng = np.random.default_rng(n_fft + m)
    wav = R.normalize((rng.standard_normal((b, c * length)) * 0.3).astype(np.float32)).reshape(b, c, length)
    plan = FrontendPlan(n_fft, hop, m, sr, c, b, length, dev)
    hip = plan.wav_to_logmel(torch.from_numpy(wav).to(dev), minmax=False, log=False).cpu().numpy()
    ref64 = R.wav_to_mel(wav, n_fft, hop, m, sr, dtype=np.float64)
    print(f"mfma-test input n_fft {n_fft} M {m} C {c}: fp32 HIP vs fp64 {rel_err(hip, ref64):.2e} | numpy fp32 vs fp64 {rel_err(R.wav_to_mel(wav, n_fft, hop, m, sr), ref64):.2e}")
