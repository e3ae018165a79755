"""Long run of the training step (eager and as a replayed graph): device memory must be flat after the first steps, losses finite.
usage: python3 scripts/gpu_leak_check.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', '64'])
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 64, 130816, dev, training=True, device_draw=True, seed=1)
gen = torch.Generator(device=dev).manual_seed(2)
wavs = [torch.randn(64, 1, 130816, generator=gen, device=dev) * 0.1 for _ in range(4)]
ys = [(torch.rand(64, 16, 3, generator=gen, device=dev) < 0.1).float() for _ in range(4)]
ok = True
for form in ("eager", "hipgraph"):
    torch.manual_seed(0)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    model.compile(S.make_optimizer(cfg, model.parameters(), capturable=(form == "hipgraph")), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    step = model.train_step if form == "eager" else S.GraphedTrainStep(model, (fe(wavs[0]), ys[0]), preserve_state=True)
    marks = []
    for s in range(steps):
        loss = step((fe(wavs[s % 4]), ys[s % 4]))['loss']
        if s % (steps // 6) == 0 or s == steps - 1:
            torch.cuda.synchronize()
            marks.append((s, torch.cuda.memory_allocated(dev) >> 20, torch.cuda.memory_reserved(dev) >> 20, float(loss)))
    print(form, " | ".join(f"step {s}: {a} MiB allocated, {r} MiB reserved, loss {l:.4f}" for s, a, r, l in marks), flush=True)
    flat = marks[-1][1] <= marks[1][1] + 64 and marks[-1][2] <= marks[1][2] + 256
    fin = all(l == l and l < 10 for _, _, _, l in marks)
    print(form, "memory flat:", flat, "losses finite:", fin)
    ok = ok and flat and fin
    del step, model
print("leak check:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
