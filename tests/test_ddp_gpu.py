"""world_size-2 data-parallel test ON THE GPU (two ranks sharing cuda:0, gloo transport - a one-GPU box has no second device
for RCCL): DistributedDataParallel around the model with every HIP pass on (first-layer recompute, BatchNorm / ReLU / MaxPool
passes, LSTM launches, fused AGC): the all-reduced gradients equal the mean of the per-rank gradients and both ranks hold
identical parameters after a full step."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from challenge_amd import sj_train as S
    S.configure_miopen()
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1'])
    torch.manual_seed(0)                      # identical init on every rank
    model = S.get_model(cfg).to(device).to(memory_format=torch.channels_last)
    ref = S.get_model(cfg).to(device).to(memory_format=torch.channels_last)
    ref.load_state_dict(model.state_dict())
    g = torch.Generator().manual_seed(100)    # the same global batch everywhere; each rank takes its shard
    xs = torch.randn(8, 32, 64, 1, generator=g).to(device)
    ys = (torch.rand(8, 2, 3, generator=g) > 0.8).float().to(device)
    x, y = xs[rank::world].contiguous(), ys[rank::world].contiguous()
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue,
                  ddp=S.wrap_ddp(model, device, world))
    grads = []                                # reference: mean over ranks of single-process gradients (BatchNorm per replica)
    for rr in range(world):
        ref.zero_grad()
        ref.train()
        S.binary_crossentropy(ys[rr::world].contiguous(), ref(xs[rr::world].contiguous())).backward()
        grads.append([p.grad.clone() for p in ref.parameters()])
    mean_grads = [sum(gs) / world for gs in zip(*grads)]
    model.train()
    model.optimizer.zero_grad()
    out = model._call(x)
    S.binary_crossentropy(y, out).backward()
    names = [n for n, _ in model.named_parameters()]
    err = max(float((p.grad - g_).abs().max()) / (float(g_.abs().max()) + 1e-6) for p, g_ in zip(model.parameters(), mean_grads))
    assert err < 1e-4, err
    fused = [type(m).__name__ for m in model.modules()]
    model.train_step((x, y))                 # full step: fused AGC + clipvalue + Adam on the averaged gradients
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    torch.save({"ok": True, "err": err, "n": len(names), "grad_fn": out.grad_fn.name() if out.grad_fn else ""},
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_sharing_the_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(tmp_path / f"rank{r}.pt")
        assert res["ok"] and res["err"] < 1e-4
