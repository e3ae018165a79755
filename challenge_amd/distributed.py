"""Data parallelism for the training step - absent in the reference (sj_train.py:408 upstream only sets CUDA_VISIBLE_DEVICES,
:513-519 is a plain `fit`): one process per GPU, torch.distributed over RCCL ('nccl' on ROCm) / xGMI, ONE exchange per step (the
bucketed gradient all-reduce), BatchNorm statistics averaged once per epoch."""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from . import switches as SW


def distributed_env(env=None):
    """Environment a multi-process GPU job needs on this stack, set BEFORE the first GPU call of the process (or in the
    environment handed to the ranks): the host driver only supports dmabuf IPC, and without HSA_ENABLE_IPC_MODE_LEGACY=0
    RCCL's intra-node transports fail with `hipIpcGetMemHandle: invalid argument`.  `bench.self_launch`, `init_distributed`
    and INTEGRATION.md's launch line all go through here, so the three cannot drift apart.  Values the user has set win."""
    env = os.environ if env is None else env
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')  # single node; the container's hostname may not resolve
    return env


def force_process_group() -> bool:
    """IRIS_FORCE_PG=1: create the process group, wrap the model in DistributedDataParallel and run every collective of
    `fit` even at world size 1.  A one-GPU box then executes the REAL backend (RCCL: communicator initialisation under
    HSA_ENABLE_IPC_MODE_LEGACY=0, DDP's reducer on RCCL's stream next to the raw-pointer HIP passes on torch's current
    stream, the all-reduce kernels themselves) - what a gloo run with two ranks sharing the device cannot show."""
    return os.environ.get('IRIS_FORCE_PG', '0') == '1'


def collectives_on(world: int) -> bool:
    """Do `fit` / `average_bn_statistics` / the bench issue their collectives?  world > 1, or a forced group at world 1."""
    return world > 1 or (force_process_group() and torch.distributed.is_available() and torch.distributed.is_initialized())


def init_distributed(force_group: Optional[bool] = None):
    """One process per GPU (torchrun): returns (rank, world, device).  Backend 'nccl' is
    RCCL on ROCm; 'gloo' on CPU-only hosts (tests).  `force_group` (default: IRIS_FORCE_PG=1) creates the group at world
    size 1 as well - in a fresh process, at its first GPU call, never after a re-exec."""
    distributed_env()  # before torch.cuda.is_available(): that call already initialises the HIP runtime
    if force_group:
        os.environ['IRIS_FORCE_PG'] = '1'
    force_group = force_process_group() if force_group is None else force_group
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    else:
        device = torch.device('cpu')
    if (world > 1 or force_group) and not torch.distributed.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if device.type == 'cuda':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    return rank, world, device


def wrap_ddp(model: nn.Module, device, world: int):
    if not collectives_on(world):
        return None
    from torch.nn.parallel import DistributedDataParallel as DDP
    # ONE collective per step - the bucketed gradient all-reduce: BatchNorm statistics stay per replica during the epoch
    # (the reference has no multi-GPU at all), so the per-forward buffer broadcast is switched off; `fit` averages them
    # over the ranks once per epoch, before validation and checkpointing (average_bn_statistics)
    return DDP(model, device_ids=[device.index] if device.type == 'cuda' else None,
               bucket_cap_mb=SW.DDP_BUCKET_MB, gradient_as_bucket_view=True, broadcast_buffers=False)


@torch.no_grad()
def average_bn_statistics(model: nn.Module, world: int) -> None:
    """BatchNorm running statistics are per replica under DDP (`broadcast_buffers=False`: no per-forward broadcast), each
    rank seeing 1 / world of the data.  Before validation and checkpointing they are averaged over the ranks - ONE small
    all-reduce per epoch over the 46 running_mean / running_var vectors flattened together - so that every rank validates,
    and rank 0 saves, the same model.  (The mean of per-rank variances ignores the spread of the per-rank means: the
    running averages of identically distributed shards, where that spread is O(1 / sqrt(steps)).)"""
    if not collectives_on(world):
        return
    bufs = [b for name, b in model.named_buffers() if name.endswith(('running_mean', 'running_var'))]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    torch.distributed.all_reduce(flat)
    flat /= world
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b))
        off += n
    if hasattr(model, 'bump_generation'):
        model.bump_generation()


def _gradient_buckets(params, cap_bytes: int, first_bytes: int = 1 << 20):
    """DistributedDataParallel's bucket order - parameters in REVERSE registration order (the order backward produces their
    gradients in), a small first bucket so that the first all-reduce starts early, then `cap_bytes` per bucket."""
    buckets, cur, size, cap = [], [], 0, first_bytes
    for p in reversed([p for p in params if p.requires_grad]):
        n = p.numel() * p.element_size()
        if cur and size + n > cap:
            buckets.append(cur)
            cur, size, cap = [], 0, cap_bytes
        cur.append(p)
        size += n
    if cur:
        buckets.append(cur)
    return buckets
