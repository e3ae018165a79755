"""Per-phase cycle breakdown (diag build, IRIS_ABLATE=4096) for one shape:
usage: IRIS_LIB=.../libiris_frontend_diag.so IRIS_ABLATE=4608 python gpu_phase_shape.py n_fft hop m c b length [bands]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from challenge_amd.frontend import FrontendPlan
n_fft, hop, m, c, b, length = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda", 0)
plan = FrontendPlan(n_fft, hop, m, 16000, c, b, length, dev)
wav = torch.randn(b, c, length, device=dev) * 0.1
for _ in range(5):
    out = plan.wav_to_logmel(wav, minmax=False, log=False)
torch.cuda.synchronize()
del plan
