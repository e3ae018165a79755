import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", os.environ.get("FM", "FAST"))
import torch
from torch.profiler import profile, ProfilerActivity
from challenge_amd import sj_train as S
dev = torch.device("cuda", 0)
batch = 64
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
x = torch.randn(batch, 64, 512, 1, device=dev)
y = (torch.rand(batch, 16, 3, device=dev) < 0.1).float()
for _ in range(2): model.train_step((x, y))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    model.train_step((x, y)); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
