"""Host-side profile of the eager c4 training step (cProfile over n steps, top entries by internal and cumulative time):
where the Python / launch overhead of the step goes when the GPU is no longer the bound.   usage: python3 scripts/gpu_host_profile.py [n]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
batch, length = 64, 130816
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, batch, length, dev, training=True, device_draw=True, seed=99)
gen = torch.Generator(device=dev).manual_seed(4321)
wav = torch.randn(batch, 1, length, generator=gen, device=dev) * 0.1
y = (torch.rand(batch, 16, 3, generator=gen, device=dev) < 0.1).float()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(5):
    model.train_step((fe(wav), y))
torch.cuda.synchronize()
# host time of a step with the GPU far behind (no synchronisation inside): n steps queued, then one wait
t0 = time.perf_counter()
for _ in range(n):
    model.train_step((fe(wav), y))
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue time {1e3 * t_host / n:.3f} ms per step; with the GPU {1e3 * t_all / n:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    model.train_step((fe(wav), y))
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in buf.getvalue().splitlines() if l.strip())[:9000])
