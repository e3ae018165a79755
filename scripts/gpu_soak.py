"""Soak: `fit` for a few epochs on the synthetic waveform corpus (device-side mixing + fused frontend + CRNN v9, batch 64 x 512 frames)
in three forms - replayed hipGraph with the exact-fp32 convolutions (the default), replayed hipGraph with the split-bf16 convolutions,
eager with the exact ones - from the same initial weights and the same data stream.  Every loss must be finite, the training loss
must fall, and the three forms must land in the same place (they differ in rounding only; Adam amplifies that, so the bound is loose).
usage: python3 scripts/gpu_soak.py [epochs] [steps_per_epoch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
batch = 64
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
init = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last).state_dict()
src = S.synthetic_wave_sources(2, 3, 256, n_bg=16, n_voice=64, n_noise=32, seed=1)
val_it = iter(S.make_wave_dataset(cfg, False, sources=src, device=dev, seed=99, n_fft=1024, hop=256, device_draw=True))
val = [tuple(t.clone() for t in next(val_it)) for _ in range(8)]
results = {}
for name, split, graph in (("hipgraph, exact fp32", False, True), ("hipgraph, split bf16", True, True), ("eager, exact fp32", False, False)):
    S.WINO_SPLIT_BF16 = split
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    model.load_state_dict(init)
    model.compile(S.make_optimizer(cfg, model.parameters(), capturable=graph), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    data = S.make_wave_dataset(cfg, True, sources=src, device=dev, seed=7, n_fft=1024, hop=256, device_draw=True)
    t0 = time.time()
    hist = S.fit(model, data, epochs, steps, validation_data=val, validation_steps=len(val), verbose=False, graph=graph)
    torch.cuda.synchronize()
    results[name] = hist
    print(f"{name}: {time.time() - t0:.1f} s for {epochs} x {steps} steps; " +
          "; ".join(f"epoch {h['epoch']}: loss {h['loss']:.4f} val {h['val_loss']:.4f}" for h in hist), flush=True)
S.WINO_SPLIT_BF16 = False
ok = True
for name, hist in results.items():
    fin = all(h['loss'] == h['loss'] and h['val_loss'] == h['val_loss'] and abs(h['loss']) < 1e3 for h in hist)
    falls = hist[-1]['loss'] < hist[0]['loss']
    ok = ok and fin and falls and len(hist) == epochs
    print(f"{name}: finite {fin}, training loss falls {falls} ({hist[0]['loss']:.4f} -> {hist[-1]['loss']:.4f})")
base = results["hipgraph, exact fp32"][-1]
for name, hist in results.items():
    d = abs(hist[-1]['loss'] - base['loss']) / base['loss']
    print(f"{name}: final training loss {hist[-1]['loss']:.4f} ({100 * d:.2f} % from the default form), val {hist[-1]['val_loss']:.4f}")
    ok = ok and d <= 0.1
print("soak:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
