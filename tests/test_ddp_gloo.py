"""world_size-2 data-parallel test on CPU (gloo): the N > 1 path of the training step.
Gradients all-reduced by DDP equal the mean of the per-rank gradients, and after
AGC + clipvalue + Adam both ranks hold identical parameters."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from challenge_amd import sj_train as S
    r, w, device = S.init_distributed()
    assert (r, w, device.type) == (rank, world, "cpu")
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1'])
    torch.manual_seed(0)                      # identical init on every rank
    model = S.get_model(cfg)
    ref = S.get_model(cfg)
    ref.load_state_dict(model.state_dict())
    g = torch.Generator().manual_seed(100)    # the same global batch everywhere; each rank takes its shard
    xs = torch.randn(4, 32, 64, 1, generator=g)
    ys = (torch.rand(4, 2, 3, generator=g) > 0.8).float()
    x, y = xs[rank::world], ys[rank::world]
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue,
                  ddp=S.wrap_ddp(model, device, world))
    # reference: mean over ranks of single-process gradients (BatchNorm stays per replica)
    grads = []
    for rr in range(world):
        ref.zero_grad()
        ref.train()
        S.binary_crossentropy(ys[rr::world], ref(xs[rr::world])).backward()
        grads.append([p.grad.clone() for p in ref.parameters()])
    mean_grads = [sum(gs) / world for gs in zip(*grads)]
    model.train()
    model.optimizer.zero_grad()
    S.binary_crossentropy(y, model._call(x)).backward()
    err = max(float((p.grad - g_).abs().max()) for p, g_ in zip(model.parameters(), mean_grads))
    assert err < 1e-5, err
    model.train_step((x, y))                 # full step: AGC + clipvalue + Adam on averaged grads
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    torch.save({"ok": True, "err": err}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(tmp_path / f"rank{r}.pt")["ok"]


def _fit_worker(rank, world, port, out_dir):
    """fit() under DDP with early stopping: the ranks see different data (so their local losses differ)
    and must still leave the epoch loop together (sj_train.py:495 EarlyStopping, :513-519 fit)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from challenge_amd import sj_train as S
    r, w, device = S.init_distributed()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1'])
    torch.manual_seed(0)
    model = S.get_model(cfg)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue,
                  ddp=S.wrap_ddp(model, device, world))

    def stream(seed):
        g = torch.Generator().manual_seed(seed)
        while True:
            yield torch.randn(2, 32, 64, 1, generator=g), (torch.rand(2, 2, 3, generator=g) > 0.8).float()

    class Val:  # a fresh, rank-specific validation stream each epoch: the rank-local val losses differ
        def __iter__(self):
            return stream(500 + rank)

    csv_path = os.path.join(out_dir, "log.csv")
    ckpt = os.path.join(out_dir, "best.pt")
    # lr so large that the loss cannot keep improving: early stopping fires long before `epochs`
    for g_ in model.optimizer.param_groups:
        g_['lr'] = 0.5
    hist = S.fit(model, stream(100 + rank), epochs=12, steps_per_epoch=2, validation_data=Val(), validation_steps=1,
                 scheduler=None, csv_path=csv_path, checkpoint_path=ckpt, patience=0, rank=rank, world=world,
                 verbose=False)
    # every rank made the same decisions on the same (all-reduced) numbers
    n = torch.tensor([len(hist)])
    ns = [torch.zeros_like(n) for _ in range(world)]
    torch.distributed.all_gather(ns, n)
    assert int(ns[0]) == int(ns[1]), ns
    # BatchNorm running statistics: per replica during the epoch (different data per rank), averaged over the ranks at
    # every epoch end - so after fit every rank holds, validates and (rank 0) checkpoints the same statistics
    stats = torch.cat([b.flatten().double() for n_, b in model.named_buffers() if n_.endswith(('running_mean', 'running_var'))])
    ss = [torch.zeros_like(stats) for _ in range(world)]
    torch.distributed.all_gather(ss, stats)
    assert torch.equal(ss[0], ss[1]) and float(stats.abs().sum()) > 0
    assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'  # init_distributed sets the multi-process GPU environment
    v = torch.tensor([h['val_loss'] for h in hist], dtype=torch.float64)
    vs = [torch.zeros_like(v) for _ in range(world)]
    torch.distributed.all_gather(vs, v)
    assert torch.equal(vs[0], vs[1])
    torch.save({"ok": True, "epochs": len(hist)}, os.path.join(out_dir, f"fit{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_fit_early_stopping_two_ranks_gloo(tmp_path):
    """Round-1 defect: rank 0 alone updated the early-stopping counter and left the loop; the other
    ranks blocked in the next all-reduce.  All ranks must return, after the same number of epochs."""
    port = _free_port()
    mp.spawn(_fit_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(tmp_path / f"fit{r}.pt") for r in range(2)]
    assert all(r["ok"] for r in res) and res[0]["epochs"] == res[1]["epochs"]
    assert res[0]["epochs"] < 12, "early stopping never fired"
    import csv as _csv
    with open(tmp_path / "log.csv") as f:
        rows = list(_csv.DictReader(f))
    assert len(rows) == res[0]["epochs"] and (tmp_path / "best.pt").exists()


def _bn_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from challenge_amd import sj_train as S
    S.init_distributed()
    bn = torch.nn.BatchNorm1d(5)
    bn.running_mean.fill_(float(rank + 1))
    bn.running_var.copy_(torch.arange(5.0) + 10 * rank)
    bn.num_batches_tracked.fill_(7 + rank)
    S.average_bn_statistics(bn, world)
    ok = (torch.allclose(bn.running_mean, torch.full((5,), 1.5)) and torch.allclose(bn.running_var, torch.arange(5.0) + 5.0)
          and int(bn.num_batches_tracked) == 7 + rank)
    torch.save({"ok": bool(ok)}, os.path.join(out_dir, f"bn{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_average_bn_statistics_two_ranks_gloo(tmp_path):
    """running_mean / running_var become the mean over the ranks (one flattened all-reduce); counters are left alone."""
    port = _free_port()
    mp.spawn(_bn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(tmp_path / f"bn{r}.pt")["ok"] for r in range(2))


def _forced_world1_worker(rank, port, out_dir):
    """IRIS_FORCE_PG=1: the process group, DDP and every collective of `fit` at world size 1 (the CPU twin of
    tests/test_ddp_gpu.py::test_ddp_rccl_world1, gloo instead of RCCL)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("IRIS_FORCE_PG", None)
    torch.set_num_threads(2)
    import torch.distributed as dist
    from challenge_amd import sj_train as S
    assert not S.collectives_on(1)
    r, w, device = S.init_distributed(force_group=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1 and S.collectives_on(1)
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1'])
    torch.manual_seed(0)
    model, plain = S.get_model(cfg), S.get_model(cfg)
    plain.load_state_dict(model.state_dict())
    ddp = S.wrap_ddp(model, device, w)
    assert ddp is not None
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue, ddp=ddp)
    plain.compile(S.make_optimizer(cfg, plain.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    g = torch.Generator().manual_seed(3)
    data = [(torch.randn(2, 32, 64, 1, generator=g), (torch.rand(2, 2, 3, generator=g) > 0.8).float()) for _ in range(3)]
    for d in data:
        model.train_step(d)
        plain.train_step(d)
    err = max(float((p - q).abs().max()) for p, q in zip(model.parameters(), plain.parameters()))
    assert err <= 1e-6, err
    calls = {"n": 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.all_reduce = counted

    def forever():
        while True:
            yield from data
    hist = S.fit(model, forever(), epochs=2, steps_per_epoch=1, validation_data=forever(), validation_steps=1, rank=0, world=1,
                 verbose=False)
    dist.all_reduce = real
    assert len(hist) == 2 and calls["n"] == 6, (len(hist), calls)  # per epoch: loss + status, BatchNorm statistics, val loss
    torch.save({"ok": True, "err": err}, os.path.join(out_dir, "forced.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_forced_process_group_at_world_size_1(tmp_path):
    mp.spawn(_forced_world1_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert torch.load(tmp_path / "forced.pt")["ok"]


def _plan_failure_worker(rank, world, port, out_dir):
    """A failed fused-epilogue wait on ONE rank (simulated: check_plans raises on rank 1 only) must stop EVERY rank with
    EpilogueTimeout after the epoch's all-reduce instead of leaving rank 0 waiting in the next collective."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from challenge_amd import _native as N
    from challenge_amd import sj_train as S
    S.init_distributed()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1'])
    torch.manual_seed(0)
    model = S.get_model(cfg)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue,
                  ddp=S.wrap_ddp(model, torch.device("cpu"), world))

    real_check = S._fe.check_plans

    def check(device=None):
        if rank == 1:
            raise N.EpilogueTimeout("plan on rank 1 gave up")
    S._fe.check_plans = check

    def stream():
        g = torch.Generator().manual_seed(7 + rank)
        while True:
            yield torch.randn(2, 32, 64, 1, generator=g), (torch.rand(2, 2, 3, generator=g) > 0.8).float()
    S._PLAN_CHECK_ON_CPU = True  # test hook: consult the plans although the loss lives on the CPU
    raised = None
    try:
        S.fit(model, stream(), epochs=3, steps_per_epoch=1, rank=rank, world=world, verbose=False)
    except N.EpilogueTimeout as exc:
        raised = str(exc)
    finally:
        S._fe.check_plans = real_check
    assert raised is not None and ("rank 1" in raised if rank == 1 else "other rank" in raised), raised
    torch.save({"ok": True, "msg": raised}, os.path.join(out_dir, f"fail{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_fit_plan_failure_on_one_rank_stops_all_ranks(tmp_path):
    mp.spawn(_plan_failure_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(tmp_path / f"fail{r}.pt")["ok"] for r in range(2))
