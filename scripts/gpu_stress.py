"""Edge geometries of the fused kernel against the oracle: more clips than CUs, one very long clip
(chunks of thousands of frames, big time-band bitmap), tiny clips, 3 channels, bands everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import frontend_ref as R
from challenge_amd.frontend import FrontendPlan

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
cases = [  # n_fft, hop, m, c, b, length, bands
    (512, 128, 40, 1, 700, 2000, True),      # 700 clips x 16 frames: several chunks per workgroup
    (1024, 256, 64, 1, 1, 2_000_000, True),  # one clip, 7813 frames
    (1024, 64, 64, 2, 3, 300_000, True),     # 4688 frames x 2 channels per clip
    (256, 64, 24, 3, 5, 40_000, True),
    (2048, 512, 128, 2, 300, 6000, True),
    (1024, 256, 64, 1, 257, 4000, False),
    (512, 256, 80, 2, 1, 300, True),         # 2 frames
]
bad = 0
for n_fft, hop, m, c, b, length, bands in cases:
    wav = (rng.standard_normal((b, c, length)) * 0.2).astype(np.float32)
    n_t, n_f = 1 + length // hop, n_fft // 2 + 1
    kw = {}
    if bands:
        kw["t_bands"] = np.stack([np.stack(R.mask_draw(rng, n_t, max(2, min(40, n_t - 1)), 4), 1) for _ in range(b)])
        kw["f_bands"] = np.stack([np.stack(R.mask_draw(rng, n_f, 16, 2), 1) for _ in range(b)])
    plan = FrontendPlan(n_fft, hop, m, 16000, c, b, length, dev)
    x = torch.from_numpy(wav).to(dev)
    out = plan.wav_to_logmel(x, minmax=False, log=False, **kw).cpu().numpy()
    ref = R.wav_to_mel(wav, n_fft, hop, m, 16000, **kw)
    e = np.abs(out - ref).max() / max(np.abs(ref).max(), 1e-6)
    full = plan.wav_to_logmel(x, **kw).cpu().numpy()
    e2 = np.abs(np.exp(full) - np.exp(R.wav_to_logmel(wav, n_fft, hop, m, 16000, **kw))).max()
    spec = plan.stft(x[: min(b, 8)]).cpu().numpy()
    fr = np.stack([R.to_ref_layout(R.stft(wav[i], n_fft, hop)) for i in range(min(b, 8))])
    e3 = np.abs(spec - fr).max() / max(np.abs(fr).max(), 1e-6)
    ok = e <= 3e-6 and e2 <= 1e-5 and e3 <= 3e-6
    bad += not ok
    print("ok  " if ok else "FAIL", (n_fft, hop, m, c, b, length, bands), f"mel {e:.1e} logmel {e2:.1e} stft {e3:.1e}", flush=True)
    del plan, x
print("failures:", bad)
sys.exit(1 if bad else 0)
