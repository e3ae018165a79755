"""Three DDP training steps over RCCL at world size 1 (IRIS_FORCE_PG=1), for a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o rccl_w1 -- python3 scripts/gpu_rccl_world1.py [batch]

The environment of a multi-process GPU job is set HERE, before torch is imported and before the first GPU call (the
profiler's preloaded library has already initialised the GPU: no env / bash -c hop after `--`).  DDP's in-place all-reduce
of a one-rank communicator may be elided by RCCL (nothing to exchange); the explicit out-of-place collectives at the end
(all_gather_into_tensor, reduce_scatter_tensor, broadcast, all_reduce of a fresh tensor) show what this stack launches."""
import json
import os
import sys

os.environ.setdefault("IRIS_FORCE_PG", "1")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import time  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from challenge_amd import sj_train as S  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S.configure_miopen()
    t0 = time.perf_counter()
    rank, world, dev = S.init_distributed()
    t_init = time.perf_counter() - t0
    assert dist.get_backend() == "nccl"
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', str(batch)])
    torch.manual_seed(0)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    ddp = S.wrap_ddp(model, dev, world)
    assert ddp is not None
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue, ddp=ddp)
    fe = S.WaveFrontend(1024, 256, 64, 16000, 1, batch, 130816, dev, training=True, device_draw=True, seed=99)
    gen = torch.Generator(device=dev).manual_seed(4321)
    wav = torch.randn(batch, 1, 130816, generator=gen, device=dev) * 0.1
    y = (torch.rand(batch, 16, 3, generator=gen, device=dev) < 0.1).float()
    for _ in range(3):
        model.train_step((fe(wav), y))
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    losses = [model.train_step((fe(wav), y))['loss'] for _ in range(3)]
    torch.cuda.synchronize(dev)
    step_ms = 1e3 * (time.perf_counter() - t0) / 3
    S.average_bn_statistics(model, world)
    # explicit collectives, out of place where the API has such a form
    x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    out = torch.empty_like(x)
    dist.all_gather_into_tensor(out, x)
    rs = torch.empty_like(x)
    dist.reduce_scatter_tensor(rs, x.clone())
    dist.broadcast(x, src=0)
    fresh = x.clone()
    dist.all_reduce(fresh)
    torch.cuda.synchronize(dev)
    ok = bool(torch.equal(out, x) and torch.equal(rs, x) and torch.equal(fresh, x))
    print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "init_process_group_s": round(t_init, 3),
                      "train_step_ms": round(step_ms, 3), "loss": [round(float(l), 5) for l in losses], "batch": batch,
                      "buckets_mb": S.DDP_BUCKET_MB, "explicit_collectives_ok": ok,
                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                      "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
