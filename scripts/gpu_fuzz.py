"""Randomised parity sweep of the frontend kernels against the NumPy oracle (not part of the test
suite: run on the GPU box when kernels change).  usage: gpu_fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import frontend_ref as R
from challenge_amd.frontend import FrontendPlan

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
bad = 0
for case in range(n_cases):
    n_fft = int(rng.choice([256, 512, 1024, 2048]))
    hop = int(rng.choice([n_fft // 4, n_fft // 2, n_fft // 8, int(rng.integers(16, n_fft)), n_fft]))
    c = int(rng.choice([1, 1, 2, 2, 3]))
    b = int(rng.integers(1, 10))
    length = int(rng.integers(n_fft // 2 + 1, 12 * n_fft))
    m = int(rng.integers(8, 161))
    sr = float(rng.choice([8000, 16000, 22050, 44100]))
    lo = float(rng.uniform(0, 300))
    hi = float(rng.uniform(max(lo + 500, sr / 8), sr / 2))
    n_t, n_f = 1 + length // hop, n_fft // 2 + 1
    use_tb, use_fb = rng.random() < 0.5, rng.random() < 0.5
    kw = {}
    try:
        if use_tb and n_t > 3:
            kw["t_bands"] = np.stack([np.stack(R.mask_draw(rng, n_t, min(8, n_t - 1), 3), 1) for _ in range(b)])
        if use_fb:
            kw["f_bands"] = np.stack([np.stack(R.mask_draw(rng, n_f, 20, 2), 1) for _ in range(b)])
        wav = (rng.standard_normal((b, c, length)) * rng.uniform(0.01, 2.0)).astype(np.float32)
        desc = f"n_fft {n_fft} hop {hop} C {c} B {b} L {length} M {m} sr {sr:.0f} [{lo:.0f},{hi:.0f}] tb {use_tb} fb {use_fb}"
        plan = FrontendPlan(n_fft, hop, m, sr, c, b, length, dev, lower_edge_hertz=lo, upper_edge_hertz=hi)
        x = torch.from_numpy(wav).to(dev)
        ref = R.wav_to_mel(wav, n_fft, hop, m, sr, lower_edge_hertz=lo, upper_edge_hertz=hi, **kw)
        out = plan.wav_to_logmel(x, minmax=False, log=False, **kw).cpu().numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        e_fused = np.abs(out - ref).max() / scale
        spec = plan.stft(x)
        full = np.stack([R.to_ref_layout(R.stft(wav[i], n_fft, hop)) for i in range(b)])
        e_stft = np.abs(spec.cpu().numpy() - full).max() / max(np.abs(full).max(), 1e-6)
        mm = plan.magmel(spec, **kw).cpu().numpy()
        e_mag = np.abs(mm - ref).max() / scale
        logm = plan.wav_to_logmel(x, **kw).cpu().numpy()
        e_log = np.abs(np.exp(logm) - np.exp(R.wav_to_logmel(wav, n_fft, hop, m, sr, lower_edge_hertz=lo, upper_edge_hertz=hi, **kw))).max()
        ok = e_fused <= 3e-6 and e_stft <= 3e-6 and e_mag <= 3e-6 and e_log <= 1e-5 and out.shape == ref.shape
        if not ok:
            bad += 1
        print(("ok  " if ok else "FAIL"), desc, f"fused {e_fused:.1e} stft {e_stft:.1e} magmel {e_mag:.1e} log {e_log:.1e}", flush=True)
        del plan
    except Exception as e:  # noqa: BLE001
        msg = str(e).split("\n")[0][:140]
        expected = isinstance(e, (ValueError, RuntimeError)) and any(k in msg for k in ("unsupported", "UNSUPPORTED", "does not fit", "must be", "band", "edge", "-2"))
        print(("skip" if expected else "EXC "), desc, type(e).__name__, msg, flush=True)
        bad += 0 if expected else 1
print("failures:", bad)
sys.exit(1 if bad else 0)
