import sys, time, torch
sys.path.insert(0, '.')
from challenge_amd.frontend import FrontendPlan, normalize
dev = torch.device('cuda', 0)
B, L = 32, 160000
gen = torch.Generator(device=dev).manual_seed(1)
wavs = [normalize(torch.randn(B, 1, L, generator=gen, device=dev)) for _ in range(20)]
for mode in ("fused", "two_kernels"):
    for ns in (1, 2, 3):
        plans = [FrontendPlan(1024, 256, 64, 16000, 1, B, L, dev) for _ in range(ns)]
        for p in plans: p.set_epilogue(mode)
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        outs = [torch.empty((B, 64, 626, 1), device=dev) for _ in range(20)]
        calls = []
        for i in range(20):
            with torch.cuda.stream(streams[i % ns]):
                calls.append(plans[i % ns].prepare(wavs[i], out=outs[i]))
        def run(n):
            for i in range(n):
                calls[i % 20].launch()
        torch.cuda.synchronize(); run(60); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(600); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 600
        st = [p.status() for p in plans]
        print(f"{mode:12s} {ns} stream(s): {1e6*dt:6.2f} us per step, status {st}", flush=True)
