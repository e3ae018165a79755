#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X feature frontend.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused hot path (STFT -> magnitude -> mel -> per-sample
min-max -> log, data resident in HBM) over BASELINE.json configs[1]: a batch of
32 x 10 s mono 16 kHz clips, n_fft 1024, hop 256, 64 mel bands.  Consecutive steps
rotate through ROTATE distinct batches (inputs AND outputs, > 256 MiB touched per cycle),
so that every step's waveform comes from HBM and not from the 256 MiB Infinity Cache.

With N > 1 there is one process per GPU: launched by the driver through
torch.distributed.run, or - when WORLD_SIZE is unset - by this script itself, which starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a fresh child process
before it touches the GPU and relays the child's JSON line and exit code.  Every rank runs
the same batch shape on its own clips (independent clips: no data-path collective, weak
scaling) and the job throughput is the sum.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, N_FFT, HOP, N_MEL = 16000, 1024, 256, 64
BATCH, SECONDS = 32, 10
ROTATE = 20  # distinct c2 batches per cycle: 20 x (20.5 MB in + 5.1 MB out) = 512 MB > 256 MiB Infinity Cache
ALGO_BYTES_PER_AUDIO_S = 4 * SR + 4 * N_MEL * SR // HOP  # fp32 wave in + fp32 mel out = 80,000
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def cpu_baseline(wav_cpu: np.ndarray):
    """Reference CPU path timed on this host (rank 0, N=1 only).  A = NumPy
    restatement (oracle, single process); B = torch.stft + torch CPU ops with all
    host threads (the engine the reference executes).  Reported value = the faster."""
    from oracle import frontend_ref as R
    from oracle.torch_cpu_ref import wav_to_logmel_cpu
    cores = os.cpu_count() or 1
    audio_s = wav_cpu.shape[0] * wav_cpu.shape[2] / SR
    # A: NumPy, whole batch, a few repetitions (~10 s budget)
    t0 = time.perf_counter()
    R.wav_to_logmel(wav_cpu[:4], N_FFT, HOP, N_MEL, SR)
    est = (time.perf_counter() - t0) * wav_cpu.shape[0] / 4
    reps_a = max(1, min(5, int(8.0 / max(est, 1e-3))))
    ta = []
    for _ in range(reps_a):
        t0 = time.perf_counter()
        R.wav_to_logmel(wav_cpu, N_FFT, HOP, N_MEL, SR)
        ta.append(time.perf_counter() - t0)
    a_rate = audio_s / float(np.median(ta))
    # B: torch CPU; small FFTs oversubscribe badly, so try a few thread counts and keep the best
    w = torch.from_numpy(R.linear_to_mel_weight_matrix(N_MEL, N_FFT // 2 + 1, SR))
    x = torch.from_numpy(wav_cpu)
    b_rate, b_threads, b_runs = 0.0, 1, 0
    for nt in sorted({1, 8, 16, 32, min(64, cores), cores}):
        if nt > cores:
            continue
        torch.set_num_threads(nt)
        for _ in range(2):
            wav_to_logmel_cpu(x, w, N_FFT, HOP)
        tb = []
        t_end = time.perf_counter() + 2.0
        while len(tb) < 5 or (time.perf_counter() < t_end and len(tb) < 100):
            t0 = time.perf_counter()
            wav_to_logmel_cpu(x, w, N_FFT, HOP)
            tb.append(time.perf_counter() - t0)
        rate = audio_s / float(np.median(tb))
        if rate > b_rate:
            b_rate, b_threads, b_runs = rate, nt, len(tb)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(max(a_rate, b_rate), 1), "unit": "audio-s/s", "cores": b_threads if b_rate >= a_rate else 1,
        "kind": "port",
        "sample": (f"same c2 batch (32 x 10 s); B=torch.stft+torch CPU ops, best of 1/8/16/32/64/{cores} threads = "
                   f"{b_threads} threads, median of {b_runs} runs = {b_rate:.0f}; A=NumPy oracle, 1 thread, median of "
                   f"{reps_a} = {a_rate:.0f}; host has {cores} hardware threads; CPU: {model}"),
        "numpy_1thread": round(a_rate, 1), "torch_best": round(b_rate, 1), "torch_best_threads": b_threads,
    }


def side_measurements(dev, rank, world, steps, fence):
    """BASELINE configs[2] and [3], reported beside the headline (never as `value`):
    c3 = fused frontend with SpecAugment + CRNN v9 forward, batch 64 x 8.176 s (T = 512);
    c4 = the full training step (frontend, forward, backward, RCCL gradient all-reduce via
    DDP when world > 1, AGC, clipvalue, Adam), batch 64 per GPU, synthetic labels."""
    import torch.distributed as dist
    from challenge_amd import sj_train as S
    S.configure_miopen()  # NORMAL find without the naive reference solvers (see sj_train.configure_miopen)
    batch, length = 64, 130816
    audio_s = batch * length / SR
    cfg = S.ARGS().get(['--v', '9', '--n_mels', str(N_MEL), '--n_frame', '512', '--n_chan', '1',
                        '--batch_size', str(batch)])
    torch.manual_seed(0)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue,
                  ddp=S.wrap_ddp(model, dev, world))
    fe = S.WaveFrontend(N_FFT, HOP, N_MEL, SR, 1, batch, length, dev, training=True, device_draw=True,
                        seed=99 + rank)
    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    wav = torch.randn(batch, 1, length, generator=gen, device=dev) * 0.1
    y = (torch.rand(batch, 16, 3, generator=gen, device=dev) < 0.1).float()

    def timed(fn, n):
        for _ in range(3):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / n

    def fwd():
        model.eval()
        with torch.no_grad():
            model(fe(wav))

    folded = S.fold_batchnorm(model)  # inference copy: eval-mode BatchNorm folded into the conv / dense in front of it

    def fwd_folded():
        with torch.no_grad():
            folded(fe(wav))

    def train():
        model.train_step((fe(wav), y))

    t_fwd = timed(fwd, steps)
    t_fwd_folded = timed(fwd_folded, steps)
    t_train = timed(train, steps)

    # input side of the reference's own training loop (spectra in, sj_train.py:74-130) at its default
    # shape: whole batches synthesised on the device (iris_mix_specs + mel kernel with bands)
    dcfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', str(batch)])
    ds = iter(S.make_device_dataset(dcfg, True, sources=S.synthetic_sources(2, 3, n_bg=16, n_voice=64, n_noise=32, seed=rank),
                                    device=dev, seed=rank))
    t_data = timed(lambda: next(ds), steps)
    del ds
    # the same from WAVEFORM corpora: mixed before the STFT (iris_mix_waves), then the fused kernel with bands
    wds = iter(S.make_wave_dataset(dcfg, True, sources=S.synthetic_wave_sources(2, 3, HOP, n_bg=16, n_voice=64, n_noise=32,
                                                                              seed=rank), device=dev, seed=rank))
    t_wdata = timed(lambda: next(wds), steps)
    del wds
    return {
        "device_dataset": {"ms_per_batch": round(1e3 * t_data, 3), "batch_per_gpu": batch,
                           "audio_s_per_s": round(world * batch * 512 * HOP / SR / t_data, 1),
                           "shape": "spectra [257, T_i, 4] resident in HBM -> log-mel [64, 80, 512, 2] + labels"},
        "wave_dataset": {"ms_per_batch": round(1e3 * t_wdata, 3), "batch_per_gpu": batch,
                         "audio_s_per_s": round(world * batch * 511 * HOP / SR / t_wdata, 1),
                         "shape": "waveforms [2, L_i] resident in HBM -> mixed [64, 2, 130816] -> log-mel [64, 80, 512, 2] + labels"},
        "c3_frontend_specaug_crnn_fwd": {"audio_s_per_s": round(world * audio_s / t_fwd, 1),
                                         "ms_per_step": round(1e3 * t_fwd, 3), "batch_per_gpu": batch,
                                         "bn_folded_for_inference": {"audio_s_per_s": round(world * audio_s / t_fwd_folded, 1),
                                                                     "ms_per_step": round(1e3 * t_fwd_folded, 3)}},
        "c4_train_step": {"audio_s_per_s": round(world * audio_s / t_train, 1), "ms_per_step": round(1e3 * t_train, 3),
                          "batch_per_gpu": batch, "n_gpus": world, "params": sum(p.numel() for p in model.parameters()),
                          "grad_allreduce": "DDP/RCCL" if world > 1 else "none"},
    }


def two_stream(dev, wavs, outs, plan_a, fence, n):
    """The c2 step issued alternately on two streams, each with its own plan (a plan's workspace belongs to one
    stream, include/iris_frontend.h).  Same rotating batches as the headline; returns whole-job throughput."""
    from challenge_amd.frontend import PipelinedFrontend
    length = wavs[0].shape[-1]
    pipe = PipelinedFrontend(2, n_fft=N_FFT, hop=HOP, n_mel=N_MEL, sample_rate=SR, channels=1, max_batch=BATCH,
                             max_len=length, device=dev)

    def run(k):
        for i in range(k):
            j = i % len(wavs)
            pipe.submit(wavs[j], wait_current=False, minmax=True, log=True, out=outs[j])  # long-lived, complete buffers
    fence()
    run(20)
    pipe.synchronize()
    fence()
    t0 = time.perf_counter()
    run(n)
    pipe.synchronize()
    fence()
    dt = (time.perf_counter() - t0) / n
    return {"us_per_step": round(1e6 * dt, 2), "audio_s_per_s": round(BATCH * SECONDS / dt, 1), "streams": 2, "steps": n}


def kernel_source_sha() -> str:
    """sha256 over the sources of the step's two kernels and of their launch geometry (the fused kernel, the FFT
    core, the min-max/log kernel, the host code that sizes chunks and grids): ties a committed PMC traffic figure to
    the code it was measured on.  The unrelated kernels of the library (mixing, STFT, spectrum -> mel, the MFMA
    variant) are left out, so that work on them does not invalidate a figure they cannot change."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "challenge_amd", "csrc")
    names = ["common.h", "host_ops.h", "host_plan.h", "iris_fft.h", "k_elementwise.h", "k_fused.h", "spectrum.h"]
    for n in names:
        with open(os.path.join(src, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def committed_traffic(kernel_key: str):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r2/pmc_traffic.json,
    written by scripts/pmc_summarise.py).  Refused (None + reason) when the kernel sources changed since."""
    path = os.path.join(ROOT, "profiles", "r2", "pmc_traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        rec = doc[kernel_key]
    except (OSError, KeyError, ValueError):
        return None, "no committed PMC pass for " + kernel_key
    sha = kernel_source_sha()
    if doc.get("kernel_src_sha") != sha:
        return None, f"PMC pass was taken at kernel sources {doc.get('kernel_src_sha')}, this build is {sha}: refused"
    return rec["hbm_bytes_per_launch"], (f"profiles/r2/pmc_traffic.json (git {doc.get('git_sha')}, kernel sources {sha}; "
                                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 on gfx950)")


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start one fresh process per GPU through torch.distributed.run and relay
    its output.  Runs BEFORE this process touches the GPU (no torch.cuda call, no HIP call): a process that has
    initialised the GPU must never exec or be replaced, so the ranks are children and we exit with their code."""
    share = os.environ.get("IRIS_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()  # counts devices without initialising them
    if not share and n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL needs it on this driver)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(proc.stdout)
    sys.exit(proc.returncode)


def batch_sweep(dev, fence, steps):
    """Fixed vs per-frame cost of the dominant kernel: one launch over B x 10 s for B = 32, 128, 512, each rotating
    through enough distinct batches to exceed the 256 MiB Infinity Cache (B = 512 touches 410 MB in ONE launch)."""
    from challenge_amd.frontend import FrontendPlan, normalize
    length = SECONDS * SR
    rows = []
    for b in (32, 128, 512):
        per_batch = b * (length * 4 + N_MEL * (1 + length // HOP) * 4)
        copies = max(2, -(-(400 << 20) // per_batch))
        plan = FrontendPlan(N_FFT, HOP, N_MEL, SR, 1, b, length, dev)
        gen = torch.Generator(device=dev).manual_seed(77 + b)
        wavs = [normalize(torch.randn(b, 1, length, generator=gen, device=dev)) for _ in range(copies)]
        outs = [torch.empty((b, N_MEL, plan.num_frames(length), 1), device=dev) for _ in range(copies)]
        for i in range(max(3, copies)):
            plan.wav_to_logmel(wavs[i % copies], out=outs[i % copies])
        plan.timing_enable(1)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            plan.wav_to_logmel(wavs[i % copies], out=outs[i % copies])
        fence()
        dt = (time.perf_counter() - t0) / steps
        n_ev, k_ms = plan.timing_read()
        plan.timing_enable(False)
        algo = ALGO_BYTES_PER_AUDIO_S * b * SECONDS
        rows.append({"batch": b, "distinct_batches": copies, "bytes_touched_per_cycle": copies * per_batch,
                     "k1_us": round(1e3 * k_ms, 2), "k1_frac_of_8TBs": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "step_us_with_event_pairs": round(1e6 * dt, 2),
                     "audio_s_per_s": round(b * SECONDS / dt, 1)})
        del wavs, outs, plan
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket the dominant kernel with HIP events (roofline.achieved = null)")
    ap.add_argument("--event-every", type=int, default=10,
                    help="event-time every n-th launch of the dominant kernel (a pair costs ~4 us of stream time: every 4th "
                         "launch measured 0.6-1.0 us per step, every 10th under 0.1)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements (batch sweep, cache-resident replay, c3 forward, c4 training step)")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--only-sweep", action="store_true", help="of the side measurements, run only the K1 batch sweep (A/B runs)")
    ap.add_argument("--resident", action="store_true",
                    help="replay ONE batch every step (Infinity-Cache resident, as round 1 measured) instead of rotating")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])  # never returns
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    # test hook: IRIS_BENCH_SHARE_GPU=1 lets several ranks share cuda:0 over gloo, to exercise the N > 1
    # control flow (self-launch, barriers, max over ranks, DDP) on a one-GPU box; never set in a real run
    share = os.environ.get("IRIS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL

    from challenge_amd.frontend import FrontendPlan, normalize

    length = SECONDS * SR
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_rot = 1 if args.resident else ROTATE
    # reference normalisation x / (10 rms), data_utils.py:32-34
    wavs = [normalize(torch.randn(BATCH, 1, length, generator=gen, device=dev, dtype=torch.float32)) for _ in range(n_rot)]
    plan = FrontendPlan(N_FFT, HOP, N_MEL, SR, 1, BATCH, length, dev)
    outs = [torch.empty((BATCH, N_MEL, plan.num_frames(length), 1), device=dev) for _ in range(n_rot)]
    cursor = [0]

    def step():
        i = cursor[0] % n_rot
        cursor[0] += 1
        plan.wav_to_logmel(wavs[i], minmax=True, log=True, out=outs[i])

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # every 10th launch of the dominant kernel carries a start/stop event pair (an event pair on every
    # launch costs ~4 us of stream time per step, which would distort `value`)
    plan.timing_enable(0 if args.no_kernel_events else args.event_every)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    n_ev, kernel_ms = plan.timing_read()
    plan.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    audio_s_per_step = BATCH * SECONDS
    value = world * audio_s_per_step * args.steps / elapsed
    result = {
        "metric": "audio-seconds/sec @16 kHz, STFT+mel frontend only (STFT+|X|+mel+min-max+log; the CRNN forward is NOT "
                  "inside `value`: see stft_mel_fwd_audio_s_per_s)",
        "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "c2: batch 32 x 10 s mono 16 kHz per GPU, n_fft 1024 hop 256 n_mel 64; "
                               "fused STFT+magnitude+mel+min-max+log (frontend only, no collective); "
                               + (f"steps rotate through {n_rot} distinct batches = "
                                  f"{n_rot * (BATCH * length * 4 + outs[0].numel() * 4) >> 20} MiB per cycle (> 256 MiB "
                                  "Infinity Cache): inputs come from HBM" if n_rot > 1 else
                                  "ONE batch replayed every step (Infinity-Cache resident)"),
                   "global_batch": world * BATCH, "parallelism": f"dp{world}"},
        "stft_mel_fwd_audio_s_per_s": None,
    }
    extras = None
    if args.only_sweep and world == 1:
        extras = {"k1_batch_sweep": batch_sweep(dev, fence, max(args.extra_steps, 20))}
    elif not args.no_extras:
        extras = side_measurements(dev, rank, world, args.extra_steps, fence)
        if world == 1:
            extras["k1_batch_sweep"] = batch_sweep(dev, fence, max(args.extra_steps, 20))
            # BASELINE configs[4]: 22.05 kHz stereo, n_fft 2048, 128 mel - banded fp32 (default) vs fp16 MFMA variant
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import gpu_c5
            extras["c5_stereo_2048_128mel"] = {"fp32_banded_default": gpu_c5.run("fp32", 40), "fp16_mfma": gpu_c5.run("fp16_mfma", 40),
                                               "default": "fp32 banded (meets north_star's 1e-5; the fp16-MFMA variant states 2e-3 and is opt-in)"}
            # two plans on two HIP streams, alternating batches: independent batches overlap (one stream's min-max/log
            # kernel and launch gaps run in the shadow of the other stream's fused kernel).  Throughput only - kernel
            # durations read under overlap include queueing, so the headline and the roofline stay single-stream.
            extras["two_stream_pipeline"] = two_stream(dev, wavs, outs, plan, fence, max(200, args.steps))
            # the round-1 configuration (one batch replayed, Infinity-Cache resident) beside the rotating one
            plan.timing_enable(1)
            fence()
            t0 = time.perf_counter()
            for _ in range(50):
                plan.wav_to_logmel(wavs[0], minmax=True, log=True, out=outs[0])
            fence()
            dt = (time.perf_counter() - t0) / 50
            n1, k1 = plan.timing_read()
            plan.timing_enable(False)
            extras["c2_cache_resident_replay"] = {"k1_us": round(1e3 * k1, 2), "step_us_with_event_pairs": round(1e6 * dt, 2)}
    if rank == 0:
        if extras:
            result["extra"] = extras
            if "c3_frontend_specaug_crnn_fwd" in extras:
                result["stft_mel_fwd_audio_s_per_s"] = extras["c3_frontend_specaug_crnn_fwd"]["audio_s_per_s"]
        algo_bytes = ALGO_BYTES_PER_AUDIO_S * audio_s_per_step  # per launch
        achieved = (algo_bytes / (kernel_ms * 1e-3) / 1e9) if n_ev and kernel_ms > 0 else None
        kernel_key = "k_wav_to_mel<10,0,false,false,1>"
        traffic, traffic_note = committed_traffic(kernel_key)
        step_gbs = algo_bytes / (elapsed / args.steps) / 1e9
        result["roofline"] = {
            "bound": "hbm", "kernel": kernel_key,
            "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
            "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": algo_bytes,
            "kernel_ms": round(kernel_ms, 5) if n_ev else None, "launches_timed": n_ev,
            # whole step (dominant kernel + min-max/log kernel + the boundary between them) against the same bytes
            "step_achieved": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
            "event_pair_floor_us": 4.1,  # begin->end of an EMPTY kernel read this way (scripts/microbench/launch_floor.hip)
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wavs[0].cpu().numpy())
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
