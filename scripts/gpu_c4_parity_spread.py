"""Spread of the c4 parity leg over inputs (fresh SpecAugment bands per call) and modes: worst gradient error and where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
from oracle import crnn_parity as P
S.configure_miopen()
dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', '64'])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 64, 130816, dev, training=True, device_draw=True, seed=99)
gen = torch.Generator(device=dev).manual_seed(4321)
wav = torch.randn(64, 1, 130816, generator=gen, device=dev) * 0.1
y = (torch.rand(64, 16, 3, generator=gen, device=dev) < 0.1).float()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    model.train_step((fe(wav), y))
for trial in range(4):
    feats = fe(wav)
    for split in (False, True):
        S.WINO_SPLIT_BF16 = split
        r = P.c4_parity(model, feats, y, clipvalue=cfg.clipvalue, unmatched=False)
        print(f"trial {trial} split {int(split)}: ok {r['ok']} grad {r['gradient_rel_worst']:.2e} @ {r['gradient_rel_worst_where']}  loss {r['loss_abs']:.1e} out {r['outputs_abs']:.1e} bn {r['bn_buffers_rel_worst']:.1e}", flush=True)
S.WINO_SPLIT_BF16 = False
