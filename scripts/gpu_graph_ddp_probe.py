"""Where does the graph-captured DDP step fail?  World size 1 over RCCL (IRIS_FORCE_PG=1), one variant per fresh child process,
stage markers on stdout, faulthandler on.  usage: gpu_graph_ddp_probe.py [variant ...]   (no argument: every variant in turn)"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["plain_with_pg", "collective_in_capture_main_thread", "hooks_async", "hooks_sync", "graphed_train_step"]


def child(variant):
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", IRIS_FORCE_PG="1")
    import torch
    import torch.distributed as dist
    from challenge_amd import sj_train as S

    def say(*a):
        print(f"[{variant}]", *a, flush=True)
    S.configure_miopen()
    rank, world, dev = S.init_distributed()
    say("process group up:", dist.get_backend(), dist.get_world_size())
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1', '--batch_size', '8'])
    torch.manual_seed(0)
    m = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(8, 32, 64, 1, device=dev)
    y = (torch.rand(8, 2, 3, device=dev) > 0.8).float()
    if variant == "collective_in_capture_main_thread":
        t = torch.ones(1 << 20, device=dev)
        for _ in range(3):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        say("capturing one all_reduce (sync form)")
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            dist.all_reduce(t)
        say("captured; replaying")
        g.replay()
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        say("capturing one all_reduce (async form + wait)")
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            w = dist.all_reduce(t, async_op=True)
            w.wait()
        g2.replay()
        torch.cuda.synchronize()
        say("ok", float(t[0]))
        return
    ddp = S.wrap_ddp(m, dev, world) if variant != "plain_with_pg" else None
    m.compile(S.make_optimizer(cfg, m.parameters(), capturable=True), S.binary_crossentropy, clipvalue=cfg.clipvalue, ddp=ddp)
    if variant in ("plain_with_pg", "graphed_train_step"):
        say("GraphedTrainStep ...")
        step = S.GraphedTrainStep(m, (x, y), preserve_state=True)
        say("captured, world", step.world)
        for _ in range(3):
            loss = float(step((x, y))['loss'])
        say("ok, loss", loss)
        return
    # hooks_* : the capture body of GraphedTrainStep re-enacted with markers
    for _ in range(3):
        m.train_step((x, y))
    torch.cuda.synchronize()
    m.optimizer.zero_grad(set_to_none=True)
    flat = torch.zeros(sum(p.numel() for p in m.parameters()), device=dev)
    # fresh leaves on the parameters' storage: their AccumulateGrad nodes are created inside the capture, on its stream (DDP's
    # reducer keeps the real parameters' nodes alive, and those belong to the default stream)
    aliases = {n: p.detach().requires_grad_(True) for n, p in m.named_parameters()}
    params = list(aliases.values())
    left = [len(params)]
    works = []

    def ready(p):
        left[0] -= 1
        if left[0] == 0:
            say("  hook: last gradient ready; copying + all_reduce from thread", __import__("threading").current_thread().name)
            off = 0
            for q in params:
                flat[off:off + q.numel()].copy_(q.grad.reshape(-1))
                off += q.numel()
            if variant == "hooks_async":
                works.append(dist.all_reduce(flat, async_op=True))
            else:
                dist.all_reduce(flat)
            say("  hook: issued")
    g = torch.cuda.CUDAGraph()
    say("capturing forward / backward with the hook")
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        handles = [p.register_post_accumulate_grad_hook(ready) for p in params]
        loss = S.binary_crossentropy(y, torch.func.functional_call(m, aliases, (x,)))
        loss.backward()
        say("  backward returned; waiting on", len(works), "works")
        for w in works:
            w.wait()
        say("  waited")
    for h in handles:
        h.remove()
    say("captured; replaying")
    g.replay()
    torch.cuda.synchronize()
    say("ok", float(loss))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    for v in (sys.argv[1:] or VARIANTS):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v], timeout=280)
        print(f"=== {v}: exit code {r.returncode}", flush=True)
