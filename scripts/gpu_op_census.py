"""Every ATen operator one eager training step dispatches (count, where from): finds launches nobody asked for - e.g. the 13 zero
"gradients" autograd materialised for the convolutions' non-differentiable statistics output (round 6).
usage: python3 scripts/gpu_op_census.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', '64'])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 64, 130816, dev, training=True, device_draw=True, seed=99)
wav = torch.randn(64, 1, 130816, device=dev) * 0.1
y = (torch.rand(64, 16, 3, device=dev) < 0.1).float()
for _ in range(3):
    model.train_step((fe(wav), y))
seen = collections.Counter()
VIEWS = ("view", "reshape", "permute", "transpose", "detach", "alias", "as_strided", "unsqueeze", "squeeze", "expand", "slice", "select", "t.default",
         "_unsafe_view", "is_", "size", "stride", "numel", "sym_", "storage_offset", "dim", "is_contiguous", "_local_scalar", "lift_fresh", "unbind", "split")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(v in name for v in VIEWS):
            st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack() if "challenge_amd" in f.filename]
            seen[(name, st[-1] if st else "(autograd / torch)")] += 1
        return out


with Spy():
    model.train_step((fe(wav), y))
torch.cuda.synchronize()
print(sum(seen.values()), "operator calls that can launch something")
for k, v in sorted(seen.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{v:4d}  {k[0]:50s} {k[1]}")
