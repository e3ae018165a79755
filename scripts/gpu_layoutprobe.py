"""CRNN v9 forward / training step per memory format and MIOpen find mode (batch 64, 64 mel x 512 frames).
usage: python scripts/gpu_layoutprobe.py [nhwc|nchw]   (MIOPEN_FIND_MODE from the environment, default NORMAL)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
fmt = sys.argv[1] if len(sys.argv) > 1 else "nhwc"
dev = torch.device("cuda", 0)
batch = 64
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev)
if fmt == "nhwc":
    model = model.to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
x = torch.randn(batch, 64, 512, 1, device=dev)
y = (torch.rand(batch, 16, 3, device=dev) < 0.1).float()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd():
    model.eval()
    with torch.no_grad():
        model(x)


print(f"{fmt} find={os.environ.get('MIOPEN_FIND_MODE')}: fwd {timed(fwd):.3f} ms, train step {timed(lambda: model.train_step((x, y))):.3f} ms", flush=True)
