"""End-to-end training rate INCLUDING the data: batches synthesised on the device (waveform-domain corpus -> mix -> fused
frontend, or spectrum-domain corpus -> mix -> mel) feeding train_step, draws on the device or on the host.
usage: python3 scripts/gpu_e2e_train.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
batch = 64
for kind in ("wave", "spectrum"):
    for dd in (True, False):
        n_chan = 1 if kind == "wave" else 2
        cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', str(n_chan), '--batch_size', str(batch)])
        torch.manual_seed(0)
        model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
        model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
        if kind == "wave":
            src = S.synthetic_wave_sources(2, 3, 256, n_bg=16, n_voice=64, n_noise=32, seed=1)
            data = iter(S.make_wave_dataset(cfg, True, sources=src, device=dev, seed=7, n_fft=1024, hop=256, device_draw=dd))
        else:
            src = S.synthetic_sources(2, 3, n_bg=16, n_voice=64, n_noise=32, seed=1)
            data = iter(S.make_device_dataset(cfg, True, sources=src, device=dev, seed=7, device_draw=dd))
        fixed = next(data)
        for _ in range(5):
            model.train_step(next(data))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.train_step(fixed)
        torch.cuda.synchronize()
        t_fixed = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            model.train_step(next(data))
        torch.cuda.synchronize()
        t_data = (time.perf_counter() - t0) / steps
        print(f"{kind:8s} corpus, draws on the {'device' if dd else 'host  '}: {1e3 * t_data:7.3f} ms per step with a fresh batch each step, "
              f"{1e3 * t_fixed:7.3f} ms on a fixed batch (x {tuple(fixed[0].shape)})")
