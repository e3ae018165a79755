#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X feature frontend.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused hot path (STFT -> magnitude -> mel -> per-sample
min-max -> log, data resident in HBM) over BASELINE.json configs[1]: a batch of
32 x 10 s mono 16 kHz clips, n_fft 1024, hop 256, 64 mel bands.  Consecutive steps
rotate through ROTATE distinct batches (inputs AND outputs, > 256 MiB touched per cycle),
so that every step's waveform comes from HBM and not from the 256 MiB Infinity Cache.

Order of a run: an untimed pre-conditioning burst (clock / cache state of a busy GPU), the W
warm-up steps, the K timed steps (no event pairs, nothing but the steps on the stream), then -
untimed and separate - the kernel-timing pass: KERNEL_PASS more rotating steps with a HIP
event pair around each kernel on the launch stream, from which `roofline` is computed.

With N > 1 there is one process per GPU: launched by the driver through
torch.distributed.run, or - when WORLD_SIZE is unset - by this script itself, which starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a fresh child process
before it touches the GPU and relays the child's JSON line and exit code.  Every rank runs
the same batch shape on its own clips (independent clips: no data-path collective, weak
scaling; `--strong` splits a fixed global batch instead) and the job throughput is the sum.
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import hashlib
import json
import os
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
PRECONDITION = 200  # untimed steps before the warm-up: the GPU reaches the clock / cache state of a busy device
KERNEL_PASS = 100   # event-timed launches of each kernel in the separate, untimed kernel-timing pass
STRONG_GLOBAL_BATCH = 256       # --strong: c2 clips over all ranks (= 8 x 32)
STRONG_GLOBAL_TRAIN_BATCH = 512  # --strong: c4 training batch over all ranks
ALGO_BYTES_PER_AUDIO_S = 4 * SR + 4 * N_MEL * SR // HOP  # fp32 wave in + fp32 mel out = 80,000
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
PMC_JSON = os.path.join("profiles", "r6", "pmc_traffic.json")


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
        ref_logmel = R.wav_to_logmel(wav_cpu, N_FFT, HOP, N_MEL, SR)
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
    }, ref_logmel


def parity(wav_cpu: np.ndarray, gpu_logmel: np.ndarray, gpu_mel: np.ndarray, ref_logmel: np.ndarray):
    """The checker leg ("mel parity check on the same run"): what the timed kernel wrote for batch 0 of the
    rotation against the oracle on the same waveforms, plus one launch without min-max / log for the mel magnitudes.
    `mel_rule_ratio` <= 1 is the stated tolerance (oracle.frontend_ref.mel_tolerance: 1e-5 |ref| + 4 eps xrms sum W against
    the fp64 oracle); `mel_rel_err_floor_1e-3` is SURVEY 8(d)'s literal metric max |d| / max(|ref|, 1e-3), reported beside it
    (torch.stft fp32, the engine the reference runs, reads 1.1e-5 on this config by that metric: profiles/r4/hip_vs_fp64_sweep.log)."""
    from oracle import frontend_ref as R
    ref, tol = R.mel_tolerance(wav_cpu, N_FFT, HOP, N_MEL, SR)
    ratio = R.mel_err_ratio(gpu_mel, ref, tol)
    d = np.abs(gpu_mel.astype(np.float64) - ref)
    floor = float((d / np.maximum(np.abs(ref), 1e-3)).max())
    strict, covered = R.mel_strict_rel_err(gpu_mel, ref, tol)  # north_star's literal 1e-5 on every ordinary element
    logabs = float(np.abs(np.exp(gpu_logmel.astype(np.float64)) - np.exp(ref_logmel.astype(np.float64))).max())
    ok = bool(ratio <= 1.0 and strict <= 1e-5 and logabs <= 5e-6)
    return {"checked": f"c2 batch 0 of the rotation, {wav_cpu.shape[0]} x {wav_cpu.shape[2] // SR} s: the timed kernel's own output "
                       "(log-mel) + one launch without min-max / log (mel) vs the oracle on the same waveforms",
            "ok": ok, "mel_rule_ratio": round(ratio, 4), "mel_rel_err_floor_1e-3": float(f"{floor:.3e}"),
            "mel_strict_rel_err": float(f"{strict:.3e}"), "mel_strict_covers": round(covered, 5),
            "logmel_exp_abs": float(f"{logabs:.3e}"),
            "bounds": {"mel_rule_ratio": 1.0, "mel_strict_rel_err": 1e-5, "logmel_exp_abs": 5e-6},
            "strict": "|mel - ref| / |ref| over the elements whose relative term dominates the rule (|ref| >= 0.048 xrms sum W)",
            "rule": "|mel - ref_fp64| <= 1e-5 |ref| + 4 eps_fp32 xrms[b,t,c] sum_k W[k,m] (oracle.frontend_ref.mel_tolerance)"}


# what `side_measurements` has finished so far: if a LATER leg hangs (a rank stuck inside a collective: the one thing a one-GPU box
# cannot rehearse is RCCL inside a captured graph at world > 1) the watchdog prints the line with these instead of nothing
PARTIAL_EXTRAS = {}


def side_measurements(dev, rank, world, steps, fence, strong=False):
    """BASELINE configs[2] and [3], reported beside the headline (never as `value`):
    c3 = fused frontend with SpecAugment + CRNN v9 forward, batch 64 x 8.176 s (T = 512);
    c4 = the full training step (frontend, forward, backward, RCCL gradient all-reduce via
    DDP when world > 1, AGC, clipvalue, Adam), batch 64 per GPU (--strong: 512 / world), synthetic labels."""
    import torch.distributed as dist
    from challenge_amd import sj_train as S
    S.configure_miopen()  # NORMAL find without the naive reference solvers (see sj_train.configure_miopen)
    batch, length = (STRONG_GLOBAL_TRAIN_BATCH // world if strong else 64), 130816
    audio_s = batch * length / SR
    cfg = S.ARGS().get(['--v', '9', '--n_mels', str(N_MEL), '--n_frame', '512', '--n_chan', '1',
                        '--batch_size', str(batch)])
    torch.manual_seed(0)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    ddp = S.wrap_ddp(model, dev, world)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue, ddp=ddp)
    fe = S.WaveFrontend(N_FFT, HOP, N_MEL, SR, 1, batch, length, dev, training=True, device_draw=True,
                        seed=99 + rank)
    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    wav = torch.randn(batch, 1, length, generator=gen, device=dev) * 0.1
    y = (torch.rand(batch, 16, 3, generator=gen, device=dev) < 0.1).float()

    host_ms = {}   # host time per step of the last `timed(..., tag=...)` calls: issue time of n steps, before the fence

    def timed(fn, n, tag=None):
        for _ in range(3):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_issue = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if S.collectives_on(world):
            t = torch.tensor([dt, t_issue], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, t_issue = float(t[0].item()), float(t[1].item())
        if tag:
            host_ms[tag] = round(1e3 * t_issue / n, 3)
        return dt / n

    def checker_leg(fn):
        """A parity leg of the side measurements: never takes the line down, but a leg that fails - or cannot run - makes the
        process exit non-zero (main)."""
        try:
            from oracle import crnn_parity as P
            return fn(P)
        except Exception as exc:
            import traceback
            traceback.print_exc()
            return {"ok": False, "error": repr(exc)[:300]}

    def fwd():
        model.eval()
        with torch.no_grad():
            model(fe(wav))

    infer = S.InferenceEngine(model, fe, wav)  # BN folded, conv + bias + ReLU fused, frontend + forward as one hipGraph

    def train():
        model.train_step((fe(wav), y))

    t_fwd = timed(fwd, steps)
    t_fwd_folded = timed(infer.eager, steps)
    t_fwd_graph = timed(infer.replay, steps) if infer.graph_ok else None
    # checker leg for c3 (before the training steps below move the weights the engine was folded from): what one more replay
    # of the TIMED graph returns - fresh SpecAugment bands, drawn on the device - against oracle.crnn_ref.RefCRNN on the same
    # features (the frontend half of c3 is covered by `parity` above and tests/test_transforms_gpu.py at this size)
    def c3_leg(P):
        rep = infer.replay().clone() if infer.graph_ok else None       # leaves the bands it drew in infer._tb / _fb
        feats = fe.plan.wav_to_logmel(wav, minmax=fe.do_minmax, log=True, t_bands=infer._tb, f_bands=infer._fb)
        return P.c3_parity(model, infer, feats, replay_out=rep)
    c3_parity = checker_leg(c3_leg)
    t_train = timed(train, steps, tag='eager')

    # where the training step goes: device time per phase from events on the stream (one extra pass, untimed)
    phases = {}
    for _ in range(max(3, steps // 2)):
        marks = []
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        x = fe(wav)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        model.train_step((x, y), _mark=mark)
        torch.cuda.synchronize(dev)
        prev, seq = e0, [("frontend", e1)] + marks
        for name, e in seq:
            phases.setdefault(name, []).append(prev.elapsed_time(e))
            prev = e
    breakdown = {k + "_ms": round(float(np.median(v)), 3) for k, v in phases.items()}

    # exposed gradient all-reduce: the same step with the collectives switched off (DDP.no_sync), A/B
    comm = None
    if ddp is not None:
        def train_nosync():
            with ddp.no_sync():
                model.train_step((fe(wav), y))
        t_nosync = timed(train_nosync, steps)
        comm = {"step_ms_with_allreduce": round(1e3 * t_train, 3), "step_ms_no_sync": round(1e3 * t_nosync, 3),
                "exposed_allreduce_ms_per_step": round(1e3 * (t_train - t_nosync), 3),
                "grad_bytes": 4 * sum(p.numel() for p in model.parameters()), "bucket_cap_mb": S.DDP_BUCKET_MB}

    PARTIAL_EXTRAS["c4_train_step"] = {
        "audio_s_per_s": round(world * audio_s / t_train, 1), "ms_per_step": round(1e3 * t_train, 3), "batch_per_gpu": batch, "n_gpus": world,
        "host_ms_per_step": host_ms.get('eager'), "allreduce": comm,
        "grad_allreduce": ("DDP/" + ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend())) if ddp is not None else "none",
        "partial": "the eager step only: a later side measurement did not finish (see `error`)"}
    PARTIAL_EXTRAS["c3_frontend_specaug_crnn_fwd"] = {"ms_per_step": round(1e3 * t_fwd, 3), "parity": c3_parity,
                                                      "inference_engine_ms_per_step": round(1e3 * t_fwd_folded, 3), "partial": True}
    PARTIAL_EXTRAS["c3_best_fp32_audio_s_per_s"] = round(world * audio_s / min(t for t in (t_fwd, t_fwd_folded, t_fwd_graph) if t is not None), 1)

    # opt-in bf16 autocast variant of the forward and the training step, with its deviation from fp32 stated
    bf16 = None
    try:
        model.eval()
        with torch.no_grad():
            feats = fe(wav)
            ref = model(feats)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                lo = model(feats).float()
        dev_abs = float((lo - ref).abs().max())

        def fwd_bf16():
            model.eval()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                model(fe(wav))

        def train_bf16():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                model.train_step((fe(wav), y))
        t_fb = timed(fwd_bf16, steps)
        t_tb = timed(train_bf16, steps)
        bf16 = {"fwd_ms_per_step": round(1e3 * t_fb, 3), "fwd_audio_s_per_s": round(world * audio_s / t_fb, 1),
                "train_ms_per_step": round(1e3 * t_tb, 3), "train_audio_s_per_s": round(world * audio_s / t_tb, 1),
                "max_abs_dev_of_sigmoid_outputs_vs_fp32": round(dev_abs, 6),
                "note": "NOT a bf16 path of this library: a smoke run of torch.autocast(bf16) around the CRNN with the STOCK torch / "
                        "MIOpen ops - every HIP pass (BatchNorm / ReLU / MaxPool / LSTM / first layer) is fp32 only and bypassed "
                        "under autocast, the frontend stays fp32.  The reference trains in fp32: fp32 is the default and the "
                        "reported metric; this entry only shows that autocast does not break the step"}
    except Exception as exc:  # an opt-in extra must never take the bench line down
        bf16 = {"error": repr(exc)[:200]}

    # the same training step as ONE replayed hipGraph (single GPU): no zero fills, no gradient-accumulate launches, no gaps
    # ... under DDP over RCCL too (round 6): the bucketed gradient all-reduce is issued from inside the capture, so with N ranks on
    # one host the step costs two copies and ONE graph launch of host time instead of ~260 kernel launches (what `fit` runs by default)
    graphed = None
    if world == 1 or (ddp is not None and dist.get_backend() == "nccl"):
        try:
            gm = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
            gm.load_state_dict(model.state_dict())
            gm.compile(S.make_optimizer(cfg, gm.parameters(), capturable=True), S.binary_crossentropy, clipvalue=cfg.clipvalue,
                       ddp=S.wrap_ddp(gm, dev, world))
            gstep = S.GraphedTrainStep(gm, (fe(wav), y))
            t_graph = timed(lambda: gstep((fe(wav), y)), steps, tag='hipgraph')
            graphed = {"ms_per_step": round(1e3 * t_graph, 3), "audio_s_per_s": round(world * audio_s / t_graph, 1),
                       "host_ms_per_step": host_ms.get('hipgraph'), "gradient_allreduce_in_graph": gstep.world or None,
                       "what": "sj_train.GraphedTrainStep: forward, loss, backward, (under DDP: the bucketed RCCL all-reduce,) AGC + "
                               "clipvalue, Adam captured once, replayed"}
            del gstep, gm
        except Exception as exc:  # an optimisation on top of the eager step: never takes the line down (every rank raises alike)
            graphed = {"error": repr(exc)[:200]}

    # checker leg for c4: one training-mode forward / backward / AGC + clipvalue on the timed batch (this step's features and
    # labels, the model where the timed steps left it), every HIP pass on, against the fp64 reference taking the same ReLU /
    # max-pool decisions (oracle/crnn_parity.py; why decisions must be matched at this size: oracle.crnn_ref.Decisions)
    c4_parity = checker_leg(lambda P: P.c4_parity(model, fe(wav), y, clipvalue=cfg.clipvalue))

    # opt-in (IRIS_WINO_SPLIT_BF16): the same c3 forward and c4 step with blocks 2-5 on the BF16 matrix cores - both operands of
    # the Winograd GEMMs split into three bf16 terms, six partial products accumulated in fp32 (k_conv_wino_b3.h) - with their own
    # parity legs under the bounds of the exact-fp32 kernels.  The headline c3 / c4 above stay on the exact-fp32 kernels.
    split = None
    try:
        was = S.WINO_SPLIT_BF16
        S.WINO_SPLIT_BF16 = True
        infer3 = S.InferenceEngine(model, fe, wav)
        t3_eager = timed(infer3.eager, steps)
        t3_graph = timed(infer3.replay, steps) if infer3.graph_ok else None

        def c3_split_leg(P):
            rep = infer3.replay().clone() if infer3.graph_ok else None
            feats = fe.plan.wav_to_logmel(wav, minmax=fe.do_minmax, log=True, t_bands=infer3._tb, f_bands=infer3._fb)
            return P.c3_parity(model, infer3, feats, replay_out=rep)
        p3 = checker_leg(c3_split_leg)
        t4 = timed(train, steps)
        g4 = None
        if world == 1:
            gm = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
            gm.load_state_dict(model.state_dict())
            gm.compile(S.make_optimizer(cfg, gm.parameters(), capturable=True), S.binary_crossentropy, clipvalue=cfg.clipvalue)
            gs = S.GraphedTrainStep(gm, (fe(wav), y))
            g4 = timed(lambda: gs((fe(wav), y)), steps)
            del gs, gm
        p4 = checker_leg(lambda P: P.c4_parity(model, fe(wav), y, clipvalue=cfg.clipvalue, unmatched=False))
        best3 = min(t for t in (t3_eager, t3_graph) if t is not None)
        split = {"what": "blocks 2-5 (12 layers forward; in training forward + backward-data) as Winograd F(2x2, 3x3) with the GEMMs on "
                         "v_mfma_f32_32x32x16_bf16: every fp32 operand = the exact sum of three bf16 terms, 6 of the 9 partial products "
                         "(all down to 2^-24) accumulated in fp32; error against fp64 0.65 - 1.14x the exact-fp32 kernel's per layer "
                         "(profiles/r6/wino_b3_check_and_time.log); opt-in: IRIS_WINO_SPLIT_BF16=1",
                 "c3_split_bf16": {"ms_per_step_eager": round(1e3 * t3_eager, 3),
                                   "ms_per_step_hipgraph": None if t3_graph is None else round(1e3 * t3_graph, 3),
                                   "audio_s_per_s": round(world * audio_s / best3, 1), "split_bf16_convolutions": infer3.split_bf16_convs,
                                   "parity": p3},
                 "c4_split_bf16": {"ms_per_step": round(1e3 * t4, 3), "ms_per_step_hipgraph": None if g4 is None else round(1e3 * g4, 3),
                                   "audio_s_per_s": round(world * audio_s / (min(t4, g4) if g4 else t4), 1), "parity": p4}}
        del infer3
    except Exception as exc:   # an opt-in extra never takes the line down
        import traceback
        traceback.print_exc()
        split = {"error": repr(exc)[:300]}
    finally:
        S.WINO_SPLIT_BF16 = was

    # input side of the reference's own training loop (spectra in, sj_train.py:74-130) at its default
    # shape: whole batches synthesised on the device (iris_mix_specs + mel kernel with bands)
    dbatch = 64
    dcfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', str(dbatch)])
    ssrc = S.synthetic_sources(2, 3, n_bg=16, n_voice=64, n_noise=32, seed=rank)
    ds = iter(S.make_device_dataset(dcfg, True, sources=ssrc, device=dev, seed=rank, device_draw=True))
    t_data = timed(lambda: next(ds), steps)
    del ds
    ds = iter(S.make_device_dataset(dcfg, True, sources=ssrc, device=dev, seed=rank))
    t_data_host = timed(lambda: next(ds), steps)
    del ds, ssrc
    # the same from WAVEFORM corpora: mixed before the STFT (iris_mix_waves), then the fused kernel with bands
    wsrc = S.synthetic_wave_sources(2, 3, HOP, n_bg=16, n_voice=64, n_noise=32, seed=rank)
    wds = iter(S.make_wave_dataset(dcfg, True, sources=wsrc, device=dev, seed=rank, device_draw=True))
    t_wdata = timed(lambda: next(wds), steps)
    del wds
    wds = iter(S.make_wave_dataset(dcfg, True, sources=wsrc, device=dev, seed=rank))
    t_wdata_host = timed(lambda: next(wds), steps)
    del wds, wsrc
    # what bounds c3 / c4 once the frontend is 1 % of them: the CRNN's convolutions on the fp32 matrix cores (157.3 TFLOP/s)
    conv_flops, wino_flops, train_flops, hw = 0.0, 0.0, 0.0, (N_MEL, 512)
    for blk in model.features:
        for m in blk.modules():
            if isinstance(m, torch.nn.Conv2d):
                f = 2.0 * batch * hw[0] * hw[1] * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                conv_flops += f
                # what the inference engine issues on the matrix cores: Winograd F(2x2, 3x3) needs 16 instead of 36 multiplies
                # per output tile wherever its kernel applies (8 | Cin, 64 | Cout: blocks 2-5)
                wino_flops += f / 2.25 if (infer.wino_convs and m.in_channels % 8 == 0 and m.out_channels % 64 == 0) else f
                # ... and the training step: forward, backward-data and weight gradient by the same rule per pass
                ci, co, big = m.in_channels, m.out_channels, max(m.in_channels, m.out_channels)
                fwd = S.WINO_TRAIN and ci % 8 == 0 and co % 64 == 0 and big >= S.WINO_TRAIN_MIN_C_FWD
                bwd = S.WINO_TRAIN and co % 8 == 0 and ci % 64 == 0 and big >= S.WINO_TRAIN_MIN_C_BWD
                wrw = S.WINO_TRAIN_WRW and ci % 32 == 0 and co % 32 == 0
                train_flops += (f / 2.25 if fwd else f) + (f / 2.25 if bwd else f) + (f / 2.25 if wrw else f)
        if isinstance(getattr(blk, "pool", None), torch.nn.MaxPool2d):
            hw = (-(-hw[0] // 2), -(-hw[1] // 2))
    best_fwd_ms = 1e3 * min(t for t in (t_fwd, t_fwd_folded, t_fwd_graph) if t is not None)
    mfma = {"conv_gflop_per_forward": round(conv_flops / 1e9, 1), "fp32_mfma_peak_tflops": 157.3,
            "c3_direct_bound_ms": round(1e3 * conv_flops / 157.3e12, 3),
            "c3_mfma_gflop_issued_by_the_engine": round(wino_flops / 1e9, 1),
            "c3_bound_ms": round(1e3 * wino_flops / 157.3e12, 3), "c3_frac_of_bound": round(1e3 * wino_flops / 157.3e12 / best_fwd_ms, 3),
            "c4_direct_bound_ms": round(3e3 * conv_flops / 157.3e12, 3),
            "c4_mfma_gflop_issued_by_the_step": round(train_flops / 1e9, 1),
            "c4_bound_ms": round(1e3 * train_flops / 157.3e12, 3), "c4_frac_of_bound": round(1e3 * train_flops / 157.3e12 / (1e3 * t_train), 3),
            "winograd_layers": infer.wino_convs, "winograd_in_training": bool(S.WINO_TRAIN),
            "winograd_weight_gradient": bool(S.WINO_TRAIN_WRW),
            "note": "forward = one pass over the convolutions, training = three (forward, backward-data, backward-weight); fp32 in / "
                    "fp32 accumulate MFMA, the precision the reference trains in.  c3_bound_ms / c4_bound_ms price the multiplies "
                    "really issued (Winograd F(2x2, 3x3) in blocks 2-5 - inference: every layer; training: forward, backward-data and "
                    "the weight gradient (k_conv_wino_wrw.h) - 2.25x fewer than the direct convolution, whose own bound "
                    "c3_direct_bound_ms the engine now runs BELOW); 157.3 TFLOP/s is the MFMA peak at 2.4 GHz - a bare MFMA "
                    "loop sustains ~131 on this chip (scripts/gpu_wino_bench.py ablation)"}
    c3 = {"audio_s_per_s": round(world * audio_s / t_fwd, 1), "ms_per_step": round(1e3 * t_fwd, 3), "batch_per_gpu": batch,
          "parity": c3_parity,
          "what": "training-mode model object in eval(): BatchNorm kernels, separate bias / ReLU kernels (the literal module)",
          "inference_engine": {
              "what": "same function for inference (sj_train.InferenceEngine): BatchNorm folded into the convolutions, every "
                      "convolution a HIP kernel with bias + ReLU (+ the block's 2x2 max-pool) fused - stencil (layer 1), implicit "
                      "GEMM on the fp32 MFMA (32 -> 32), Winograd F(2x2, 3x3) on the fp32 MFMA (blocks 2-5) -, fp32; outputs equal "
                      "to 1e-4 (GPU test)",
              "eager": {"audio_s_per_s": round(world * audio_s / t_fwd_folded, 1), "ms_per_step": round(1e3 * t_fwd_folded, 3)},
              "hipgraph_replay": None if t_fwd_graph is None else {
                  "audio_s_per_s": round(world * audio_s / t_fwd_graph, 1), "ms_per_step": round(1e3 * t_fwd_graph, 3)},
              "fused_conv_bias_relu": infer.fused_convs, "hip_convolutions": infer.hip_convs, "winograd_convolutions": infer.wino_convs}}
    best_fwd = min(t for t in (t_fwd, t_fwd_folded, t_fwd_graph) if t is not None)
    return {
        "device_dataset": {"ms_per_batch": round(1e3 * t_data, 3), "ms_per_batch_host_draws": round(1e3 * t_data_host, 3),
                           "draws": "on the device (iris_mix_draw + iris_augment_draw)", "batch_per_gpu": dbatch,
                           "audio_s_per_s": round(world * dbatch * 512 * HOP / SR / t_data, 1),
                           "shape": "spectra [257, T_i, 4] resident in HBM -> log-mel [64, 80, 512, 2] + labels"},
        "wave_dataset": {"ms_per_batch": round(1e3 * t_wdata, 3), "ms_per_batch_host_draws": round(1e3 * t_wdata_host, 3),
                         "draws": "on the device (iris_mix_draw + iris_augment_draw)", "batch_per_gpu": dbatch,
                         "audio_s_per_s": round(world * dbatch * 511 * HOP / SR / t_wdata, 1),
                         "shape": "waveforms [2, L_i] resident in HBM -> mixed [64, 2, 130816] -> log-mel [64, 80, 512, 2] + labels"},
        "c3_frontend_specaug_crnn_fwd": c3,
        "c3_best_fp32_audio_s_per_s": round(world * audio_s / best_fwd, 1),
        "crnn_matrix_core_bound": mfma,
        "c4_train_step": {"audio_s_per_s": round(world * audio_s / t_train, 1), "ms_per_step": round(1e3 * t_train, 3),
                          "batch_per_gpu": batch, "n_gpus": world, "params": sum(p.numel() for p in model.parameters()),
                          "parity": c4_parity,
                          "host_ms_per_step": host_ms.get('eager'), "hipgraph": graphed, "grad_allreduce": ("DDP/" + ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend())) if ddp is not None else "none", "device_ms_per_phase": breakdown,
                          "allreduce": comm},
        "split_bf16_matrix_cores": split,
        "autocast_bf16_smoke_stock_ops": bf16,
        # whether the c3 / c4 numbers above ran on the shipped, tuned MIOpen perf-db or on this build's own defaults
        "miopen_db": S.miopen_db_status(),
    }


def two_stream(dev, wavs, outs, fence, n, n_streams=3):
    """The c2 step issued round-robin on several streams, each with its own plan (a plan's workspace belongs to one
    stream, include/iris_frontend.h) in the two-kernel form, through prepared launches.  Same rotating batches as the
    headline; returns whole-job throughput.  Independent batches overlap: one stream's launch latency, prologue and tail
    run in the shadow of another's frames - throughput only, the headline and the roofline stay single-stream."""
    from challenge_amd.frontend import PipelinedFrontend
    length = wavs[0].shape[-1]
    pipe = PipelinedFrontend(n_streams, n_fft=N_FFT, hop=HOP, n_mel=N_MEL, sample_rate=SR, channels=1, max_batch=wavs[0].shape[0],
                             max_len=length, device=dev)
    calls = []
    for j in range(len(wavs)):
        with torch.cuda.stream(pipe.streams[j % n_streams]):
            calls.append(pipe.plans[j % n_streams].prepare(wavs[j], out=outs[j], minmax=True, log=True))

    def run(k):
        for i in range(k):
            calls[i % len(calls)].launch()
    fence()
    run(3 * len(calls))
    pipe.synchronize()
    fence()
    t0 = time.perf_counter()
    run(n)
    pipe.synchronize()
    fence()
    dt = (time.perf_counter() - t0) / n
    return {"us_per_step": round(1e6 * dt, 2), "audio_s_per_s": round(wavs[0].shape[0] * SECONDS / dt, 1), "streams": n_streams,
            "epilogue": "two_kernels", "steps": n}


def graph_replay(dev, plan, wavs, outs, fence, n):
    """The c2 step captured once per rotating batch into a hipGraph (FrontendPlan.capture) and replayed: what the
    host-side launch path costs when it is taken out (same device work, same rotation)."""
    graphs = [plan.capture(wavs[i], out=outs[i], minmax=True, log=True) for i in range(len(wavs))]
    for g in graphs:
        g.replay()
    fence()
    t0 = time.perf_counter()
    for i in range(n):
        graphs[i % len(graphs)].replay()
    fence()
    dt = (time.perf_counter() - t0) / n
    return {"us_per_step": round(1e6 * dt, 2), "audio_s_per_s": round(wavs[0].shape[0] * SECONDS / dt, 1), "graphs": len(graphs),
            "steps": n}


def kernel_source_sha() -> str:
    """sha256 over the sources of the step's kernels and of their launch geometry (the fused kernel, the FFT
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


def committed_traffic():
    """HBM bytes per launch of the step's kernels from the committed PMC passes (profiles/r6/pmc_traffic.json,
    written by scripts/pmc_summarise.py).  Refused (None + reason) when the kernel sources changed since.
    Returns (dominant-kernel bytes, whole-step bytes, note)."""
    path = os.path.join(ROOT, PMC_JSON)
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, None, "no committed PMC pass (" + PMC_JSON + ")"
    sha = kernel_source_sha()
    if doc.get("kernel_src_sha") != sha:
        return None, None, f"PMC pass was taken at kernel sources {doc.get('kernel_src_sha')}, this build is {sha}: refused"
    k1 = [v for k, v in doc.items() if k.startswith("k_wav_to_mel") and isinstance(v, dict) and "hbm_bytes_per_launch" in v]
    k2 = [v for k, v in doc.items() if k.startswith("k_minmax") and isinstance(v, dict) and "hbm_bytes_per_launch" in v]
    if not k1:
        return None, None, "no fused-kernel record in " + PMC_JSON
    step = k1[0]["hbm_bytes_per_launch"] + sum(v["hbm_bytes_per_launch"] for v in k2)
    return k1[0]["hbm_bytes_per_launch"], step, (
        f"{PMC_JSON} (git {doc.get('git_sha')}, kernel sources {sha}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
        "passes, FETCH_SIZE x2 on gfx950)")


def visible_gpu_count():
    """GPUs this process would see, WITHOUT touching the HIP runtime (a parent that has initialised the GPU holds a
    KFD handle for the whole run): the *_VISIBLE_DEVICES lists if set, else the KFD topology in sysfs."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(prop) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start one fresh process per GPU through torch.distributed.run and relay
    its output.  This parent never calls into HIP or torch.cuda (devices are counted from the environment / sysfs): a
    process that has initialised the GPU must never exec or be replaced, so the ranks are children and we exit with
    their code.  --standalone lets the launcher pick (and hold) a free rendezvous port on 127.0.0.1 itself."""
    share = os.environ.get("IRIS_BENCH_SHARE_GPU") == "1"
    n_dev = visible_gpu_count()
    if not share and n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + argv
    from challenge_amd.sj_train import distributed_env  # imports torch, touches no GPU
    env = distributed_env(dict(os.environ))  # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL needs it on this driver), MASTER_ADDR
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(proc.stdout)
    sys.exit(proc.returncode)


def batch_sweep(dev, fence, steps):
    """Fixed vs per-frame cost of the dominant kernel: one launch over B x 10 s for B = 32, 128, 256, 512, each rotating
    through enough distinct batches to exceed the 256 MiB Infinity Cache (B = 512 touches 410 MB in ONE launch)."""
    from challenge_amd.frontend import FrontendPlan, normalize
    length = SECONDS * SR
    rows = []
    for b in (32, 128, 256, 512):
        per_batch = b * (length * 4 + N_MEL * (1 + length // HOP) * 4)
        copies = max(2, -(-(400 << 20) // per_batch))
        plan = FrontendPlan(N_FFT, HOP, N_MEL, SR, 1, b, length, dev)
        gen = torch.Generator(device=dev).manual_seed(77 + b)
        wavs = [normalize(torch.randn(b, 1, length, generator=gen, device=dev)) for _ in range(copies)]
        outs = [torch.empty((b, N_MEL, plan.num_frames(length), 1), device=dev) for _ in range(copies)]
        for i in range(max(3, copies)):
            plan.wav_to_logmel(wavs[i % copies], out=outs[i % copies])
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            plan.wav_to_logmel(wavs[i % copies], out=outs[i % copies])
        fence()
        dt = (time.perf_counter() - t0) / steps
        plan.timing_enable(1)
        for i in range(steps + 4):
            plan.wav_to_logmel(wavs[i % copies], out=outs[i % copies])
        fence()
        k, k2 = plan.timing_samples(0), plan.timing_samples(1)
        plan.timing_enable(False)
        k_ms = float(k.mean()) if len(k) else float("nan")
        algo = ALGO_BYTES_PER_AUDIO_S * b * SECONDS
        # from B = 128 on a chunk's mel tile no longer fits the LDS at this geometry: the library takes the two-kernel form by
        # itself, and `k1_us` is then the fused kernel WITHOUT the epilogue (the second kernel is listed beside it)
        rows.append({"batch": b, "distinct_batches": copies, "bytes_touched_per_cycle": copies * per_batch,
                     "form": "two_kernels" if len(k2) else "fused_epilogue", "epilogue": plan.last_epilogue(),
                     "second_kernel_us": round(1e3 * float(k2.mean()), 2) if len(k2) else None,
                     "k1_us": round(1e3 * k_ms, 2), "k1_us_median": round(1e3 * float(np.median(k)), 2) if len(k) else None,
                     "k1_frac_of_8TBs": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "step_us": round(1e6 * dt, 2), "step_frac_of_8TBs": round(algo / dt / 1e9 / HBM_PEAK_GBS, 4),
                     "audio_s_per_s": round(b * SECONDS / dt, 1)})
        del wavs, outs, plan
        torch.cuda.empty_cache()
    return rows


def config_rooflines(dev, fence, steps):
    """The fused step at the other BASELINE shapes, each as a fraction of the HBM roofline (event-timed kernel, inputs
    rotating through > 256 MiB): c1 (one 2 s clip - launch-bound by construction), c3 with its SpecAugment / stft_filter
    bands, the reference's own default shape (n_fft 512, hop 256, 80 mel, stereo) at its default batch 12 and at 64.
    c2 is the headline; c5 has its own entry.  Algorithmic bytes = B (4 C L + 4 M T C)."""
    from challenge_amd.frontend import FrontendPlan, normalize
    rows = {}
    shapes = [("c1_1x2s", 1024, 256, 64, 16000, 1, 1, 32000, False),
              ("c3_64x8.2s_specaugment_bands", 1024, 256, 64, 16000, 1, 64, 130816, True),
              ("reference_default_n512_m80_stereo_b12", 512, 256, 80, 16000, 2, 12, 130816, False),
              ("reference_default_n512_m80_stereo_b64", 512, 256, 80, 16000, 2, 64, 130816, False)]
    for name, n_fft, hop, m, sr, c, b, length, bands in shapes:
        plan = FrontendPlan(n_fft, hop, m, sr, c, b, length, dev)
        t = plan.num_frames(length)
        algo = b * (c * length * 4 + m * t * c * 4)
        copies = max(2, min(64, -(-(300 << 20) // algo)))
        gen = torch.Generator(device=dev).manual_seed(11 + b)
        wavs = [normalize(torch.randn(b, c, length, generator=gen, device=dev)) for _ in range(copies)]
        outs = [torch.empty((b, m, t, c), device=dev) for _ in range(copies)]
        kw = {}
        if bands:  # data_utils.augment: 6 time bands of < 24 frames, 1 frequency band of < 16 linear bins (transforms.py:12-40)
            rng = np.random.default_rng(5)
            ts, fs = rng.integers(0, 24, (b, 6)), rng.integers(0, 16, (b, 1))
            tb = np.stack([rng.integers(0, t - ts), ts], -1).astype(np.int32)
            fb = np.stack([rng.integers(0, n_fft // 2 + 1 - fs), fs], -1).astype(np.int32)
            kw = {"t_bands": torch.from_numpy(tb).to(dev), "f_bands": torch.from_numpy(fb).to(dev)}
        calls = [plan.prepare(wavs[i], out=outs[i], **kw) for i in range(copies)]
        for i in range(max(copies, 8)):
            calls[i % copies].launch()
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            calls[i % copies].launch()
        fence()
        dt = (time.perf_counter() - t0) / steps
        plan.timing_enable(1)
        for i in range(steps + 4):
            calls[i % copies].launch()
        fence()
        k, k2 = plan.timing_samples(0), plan.timing_samples(1)
        plan.timing_enable(False)
        k_ms = float(k.mean()) + (float(k2.mean()) if len(k2) else 0.0)
        rows[name] = {"kernel": plan.fused_kernel_name(with_bands=bands), "form": "two_kernels" if len(k2) else "fused_epilogue",
                      "epilogue": plan.last_epilogue(),
                      "algorithmic_bytes_per_launch": algo, "kernel_us": round(1e3 * k_ms, 2), "step_us": round(1e6 * dt, 2),
                      "frac_of_8TBs": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "step_frac_of_8TBs": round(algo / dt / 1e9 / HBM_PEAK_GBS, 4),
                      "audio_s_per_s": round(b * length / sr / dt, 1)}
        del calls, wavs, outs, plan
        torch.cuda.empty_cache()
    return rows


def device_identity(dev):
    """What tells two ranks' devices apart in the JSON line: PCI bus id and uuid where torch exposes them."""
    p = torch.cuda.get_device_properties(dev)
    ident = {"name": p.name}
    for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
        if hasattr(p, k):
            ident[k] = int(getattr(p, k))
    if hasattr(p, "uuid"):
        ident["uuid"] = str(p.uuid)
    return ident


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="skip the kernel-timing pass (roofline.achieved = null)")
    ap.add_argument("--no-precondition", action="store_true", help="skip the untimed pre-conditioning burst")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements (batch sweep, cache-resident replay, c3 forward, c4 training step)")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--extras-limit", type=float, default=300.0,
                    help="seconds the side measurements may take before the line is printed without them")
    ap.add_argument("--only-sweep", action="store_true", help="of the side measurements, run only the K1 batch sweep (A/B runs)")
    ap.add_argument("--resident", action="store_true",
                    help="replay ONE batch every step (Infinity-Cache resident, as round 1 measured) instead of rotating")
    ap.add_argument("--strong", action="store_true",
                    help=f"strong scaling: a fixed global batch ({STRONG_GLOBAL_BATCH} c2 clips; {STRONG_GLOBAL_TRAIN_BATCH} for "
                         "the c4 training step) split over the ranks instead of a fixed batch per GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])  # never returns
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("IRIS_FORCE_PG") == "1":  # ranks started by the driver's own `python -m torch.distributed.run
        from challenge_amd.sj_train import distributed_env  # ... bench.py`: same environment as self_launch gives its children,
        distributed_env()                                   # before the first GPU call
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    # test hook: IRIS_BENCH_SHARE_GPU=1 lets several ranks share cuda:0 over gloo, to exercise the N > 1
    # control flow (self-launch, barriers, max over ranks, DDP) on a one-GPU box; never set in a real run
    share = os.environ.get("IRIS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    # IRIS_FORCE_PG=1 (sj_train.force_process_group): the process group, DDP and every collective of the line at world size 1
    # too - a one-GPU box then runs the REAL backend (RCCL communicator, DDP's reducer on RCCL's stream, barrier / all-reduce
    # / all_gather_object of this script) instead of skipping it
    force_pg = world == 1 and os.environ.get("IRIS_FORCE_PG") == "1"
    coll = world > 1 or force_pg
    if coll:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner to STDOUT when its first communicator comes up; this script's stdout is ONE JSON line:
        # the banner goes to stderr (fd-level redirect around the group's creation and first collective)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if share:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        backend = dist.get_backend()
        if not share and backend != "nccl":  # the N > 1 line is about RCCL over xGMI; anything else is a mis-launch
            print(f"bench.py: world {world} runs on backend '{backend}', expected 'nccl' (RCCL)", file=sys.stderr)
            sys.exit(2)
    if args.strong and STRONG_GLOBAL_BATCH % world:
        print(f"bench.py: --strong needs a world size that divides {STRONG_GLOBAL_BATCH}", file=sys.stderr)
        sys.exit(2)
    batch = STRONG_GLOBAL_BATCH // world if args.strong else BATCH

    from challenge_amd.frontend import FrontendPlan, normalize

    length = SECONDS * SR
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    per_batch = batch * (length * 4 + N_MEL * (1 + length // HOP) * 4)
    n_rot = 1 if args.resident else max(2, min(ROTATE, -(-(488 << 20) // per_batch)))
    # reference normalisation x / (10 rms), data_utils.py:32-34
    wavs = [normalize(torch.randn(batch, 1, length, generator=gen, device=dev, dtype=torch.float32)) for _ in range(n_rot)]
    plan = FrontendPlan(N_FFT, HOP, N_MEL, SR, 1, batch, length, dev)
    outs = [torch.empty((batch, N_MEL, plan.num_frames(length), 1), device=dev) for _ in range(n_rot)]
    cursor = [0]
    # prepared launches (FrontendPlan.prepare): shapes and pointers of the long-lived rotating buffers are validated once,
    # a step is then the bare C-ABI call - a 20-step window is ~0.45 ms long and must not wait for the host
    calls = [plan.prepare(wavs[i], out=outs[i], minmax=True, log=True) for i in range(n_rot)]

    def step():
        i = cursor[0] % n_rot
        cursor[0] += 1
        calls[i].launch()

    def fence():
        torch.cuda.synchronize(dev)
        if coll:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if not args.no_precondition:
        for _ in range(PRECONDITION):
            step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    if coll:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel-timing pass: separate from the timed region, independent of --steps
    k1 = k2 = np.zeros(0, np.float32)
    if not args.no_kernel_events:
        plan.timing_enable(1)  # every call; the library never samples the first 4 calls after enabling
        for _ in range(KERNEL_PASS + 4):
            step()
        fence()
        k1, k2 = plan.timing_samples(0), plan.timing_samples(1)
        plan.timing_enable(False)

    # the same step in its two-kernel form (round-2 structure: fused kernel without the epilogue + min-max / log kernel),
    # event-timed the same way: like-for-like continuity of the per-kernel figures
    two = None
    if not args.no_kernel_events and rank == 0:
        plan2 = FrontendPlan(N_FFT, HOP, N_MEL, SR, 1, batch, length, dev)
        plan2.set_epilogue("two_kernels")
        plan2.timing_enable(1)
        for i in range(54):
            plan2.wav_to_logmel(wavs[i % n_rot], minmax=True, log=True, out=outs[i % n_rot])
        torch.cuda.synchronize(dev)
        a1, a2 = plan2.timing_samples(0), plan2.timing_samples(1)
        plan2.timing_enable(False)
        if len(a1) and len(a2):
            two = {"kernel": plan2.fused_kernel_name(), "kernel_ms": round(float(a1.mean()), 5),
                   "frac": round(ALGO_BYTES_PER_AUDIO_S * batch * SECONDS / (float(a1.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "second_kernel": "k_minmax_log_apply", "second_kernel_ms": round(float(a2.mean()), 5), "launches_timed": int(len(a1))}
        del plan2

    audio_s_per_step = batch * SECONDS
    value = world * audio_s_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    result = {
        "metric": "audio-seconds/sec @16 kHz, STFT+mel frontend only (STFT+|X|+mel+min-max+log; the CRNN forward is NOT "
                  "inside `value`: see stft_mel_fwd_audio_s_per_s)",
        "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"c2: batch {batch} x 10 s mono 16 kHz per GPU, n_fft 1024 hop 256 n_mel 64; "
                               "fused STFT+magnitude+mel+min-max+log (frontend only, no collective); "
                               + (f"steps rotate through {n_rot} distinct batches = "
                                  f"{n_rot * per_batch >> 20} MiB per cycle (> 256 MiB "
                                  "Infinity Cache): inputs come from HBM" if n_rot > 1 else
                                  "ONE batch replayed every step (Infinity-Cache resident)")
                               + ("" if args.no_precondition else
                                  f"; {PRECONDITION} untimed pre-conditioning steps run before the warm-up (busy-GPU clock / cache state)"),
                   "global_batch": world * batch, "parallelism": f"dp{world}"},
        "stft_mel_fwd_audio_s_per_s": None,
    }
    if coll:  # self-audit of the N > 1 run: who ran where, and did the collective backend see N ranks
        mine = {"rank": rank, "local_rank": local_rank, "device": device_identity(dev),
                "value": round(audio_s_per_step * args.steps / elapsed_local, 1),
                "ms_per_step": round(1e3 * elapsed_local / args.steps, 5)}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        result["ranks"] = ranks
        result["rccl_world"] = dist.get_world_size()
        result["backend"] = backend
        result["backend_is_rccl"] = backend == "nccl"  # false only under the share-one-GPU test hook (gloo)
        # `value` scales trivially (no collective in the frontend): the quantity north_star's ">= 6x at 8 GPUs" is about is
        # the TRAINING STEP, lifted to the top level below once it has been measured (side_measurements)
        result["metric"] = ("audio-seconds/sec @16 kHz, STFT+mel frontend only, whole job (embarrassingly parallel: no "
                            "collective).  SCALING TARGET = the training step: see train_step_ms / train_step_audio_s_per_s / "
                            "allreduce_exposed_ms at the top level of this line (c4: frontend + CRNN forward / backward + "
                            "RCCL gradient all-reduce + AGC + Adam, batch 64 per GPU)")
        result.update({"train_step_ms": None, "train_step_audio_s_per_s": None, "train_step_form": None, "host_ms_per_step": None,
                       "allreduce_exposed_ms": None, "grad_bytes": None})
        if force_pg:
            result["forced_process_group"] = ("IRIS_FORCE_PG=1: world size 1 with the process group, DDP and every collective "
                                              "on; allreduce_exposed_ms is then the floor of the bucket launches (no peer)")
    import threading
    done_lock, done = threading.Lock(), []
    # what the final line needs from the device is fetched now: finish() may run while the GPU is stuck
    kernel_name = plan.fused_kernel_name()
    cpu_input = wavs[0].cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    gpu_logmel = gpu_mel = None
    if cpu_input is not None:  # parity leg: batch 0 as the timed kernel wrote it + its mel magnitudes (one extra launch)
        gpu_logmel = outs[0].cpu().numpy()
        gpu_mel = plan.wav_to_logmel(wavs[0], minmax=False, log=False).cpu().numpy()

    parity_ok = []

    def finish(extras):
        """Rank 0: assemble and print the ONE JSON line (once: the watchdog below may get here first)."""
        with done_lock:
            if done or rank != 0:
                return
            done.append(True)
            if extras:
                result["extra"] = extras
                if "c3_best_fp32_audio_s_per_s" in extras:
                    result["stft_mel_fwd_audio_s_per_s"] = extras["c3_best_fp32_audio_s_per_s"]
                c4 = extras.get("c4_train_step")
                if coll and c4:  # the scaling curve of the END-TO-END TRAINING STEP, at the top level of the line
                    comm = c4.get("allreduce") or {}
                    g = c4.get("hipgraph") or {}
                    use_graph = g.get("ms_per_step") is not None and g["ms_per_step"] < c4.get("ms_per_step", float("inf"))
                    form = g if use_graph else c4   # what `fit` runs: the replayed graph wherever it could be captured
                    result.update({"train_step_ms": form.get("ms_per_step"), "train_step_audio_s_per_s": form.get("audio_s_per_s"),
                                   "train_step_form": "hipgraph replay (all-reduce inside the graph)" if use_graph else "eager DDP",
                                   "host_ms_per_step": form.get("host_ms_per_step"),
                                   "train_step_eager_ms": c4.get("ms_per_step"), "train_step_eager_host_ms": c4.get("host_ms_per_step"),
                                   "allreduce_exposed_ms": comm.get("exposed_allreduce_ms_per_step"),
                                   "grad_bytes": comm.get("grad_bytes")})
            algo_bytes = ALGO_BYTES_PER_AUDIO_S * audio_s_per_step  # per launch
            traffic, step_traffic, traffic_note = committed_traffic()
            step_gbs = algo_bytes / (elapsed / args.steps) / 1e9
            roof = {
                "bound": "hbm", "kernel": kernel_name, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None, "traffic": traffic, "step_traffic": step_traffic, "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": algo_bytes,
                # whole step (every kernel of the step + the boundaries between them) against the same bytes
                "step_achieved": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
                "how": f"kernel-timing pass after the timed region: {KERNEL_PASS} rotating steps with a HIP event pair around each "
                       "kernel on the launch stream (hipExtLaunchKernel); `achieved` = algorithmic bytes / MEAN duration of the "
                       "dominant kernel.  Such a pair reads marker-end -> kernel-end on the launch stream; against rocprofv3's "
                       "duration of the same kernel under the same command the two agree within ~1 us, either way round "
                       "(profiles/r4/driver_cmd_kernel_stats.csv vs bench_driver_cmd.json of the same session)",
            }
            if len(k1):
                kernel_ms, k2_ms = float(k1.mean()), (float(k2.mean()) if len(k2) else 0.0)
                roof.update({"kernel_ms": round(kernel_ms, 5), "kernel_ms_median": round(float(np.median(k1)), 5),
                             "kernel_ms_min": round(float(k1.min()), 5), "launches_timed": int(len(k1)),
                             "second_kernel": "k_minmax_log_apply" if len(k2) else None,
                             "second_kernel_ms": round(k2_ms, 5) if len(k2) else None,
                             "second_kernel_ms_median": round(float(np.median(k2)), 5) if len(k2) else None})
                if kernel_ms + k2_ms <= ms_per_step * 1.05 + 0.004:  # the event reading carries up to ~2 us of dispatch gap per kernel
                    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
                    roof["achieved"], roof["frac"] = round(achieved, 1), round(achieved / HBM_PEAK_GBS, 4)
                else:
                    roof["frac_withheld"] = (f"kernel_ms {kernel_ms:.5f} + second_kernel_ms {k2_ms:.5f} exceed ms_per_step "
                                             f"{ms_per_step:.5f} x 1.05: inconsistent, no fraction reported")
            # the OTHER bound of this kernel: it is bound by vector-instruction issue, not by bytes.
            # 235 vector instructions per frame (PMC, profiles/r4/pmc_traffic.json) x 4.3 cycles per packed-fp32 instruction and
            # SIMD (scripts/microbench/valu_rate.hip) over 1,024 SIMDs at 2.4 GHz, plus the launch's fixed cost that no frame
            # work can hide (dispatch + drain 2.4 us, first HBM fetch of the prologue 2.5 us: in-kernel stamps, round 3 / 4)
            frames = batch * plan.channels * plan.num_frames(length)
            issue_us = frames * 235 * 4.3 / (1024 * 2.4e3)
            roof["issue_floor_us"] = round(issue_us + 4.9, 2)
            roof["issue_floor_frac"] = round(algo_bytes / ((issue_us + 4.9) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            roof["issue_floor_how"] = ("frames x 235 vector instructions x 4.3 cycles / (1,024 SIMDs x 2.4 GHz) + 4.9 us of launch, drain "
                                       "and first fetch: what THIS decomposition (one wave per frame, fp32 Stockham 8x8x8 on packed "
                                       "math) could reach with every LDS / memory wait hidden; round 5's A/B of five waves per SIMD "
                                       "against four on one instruction stream lost 19 % (profiles/r5/ab_occupancy.log)")
            roof["two_kernel_form"] = two
            result["roofline"] = roof
            if cpu_input is not None:
                result["cpu_baseline"], ref_logmel = cpu_baseline(cpu_input)
                result["parity"] = parity(cpu_input, gpu_logmel, gpu_mel, ref_logmel)
                parity_ok.append(result["parity"]["ok"])
            print(json.dumps(result), flush=True)

    # A side measurement that hangs (one rank failing inside a collective while the others wait) must not take the
    # headline down with it: after --extras-limit seconds rank 0 prints the line without the extras and every rank leaves.
    def abandon():
        print(f"bench.py: rank {rank}: side measurements exceeded {args.extras_limit} s, abandoned", file=sys.stderr, flush=True)
        finish(dict(PARTIAL_EXTRAS, error=f"side measurements exceeded {args.extras_limit} s and were abandoned; what had finished by then is kept"))
        os._exit(0 if all(parity_ok) else 3)
    timer = threading.Timer(args.extras_limit, abandon)
    timer.daemon = True
    if not args.no_extras:
        timer.start()
    extras = None
    if args.only_sweep and world == 1:
        extras = {"k1_batch_sweep": batch_sweep(dev, fence, max(args.extra_steps, 20))}
    elif not args.no_extras:
        try:
            extras = side_measurements(dev, rank, world, args.extra_steps, fence, args.strong)
        except Exception as exc:  # the side measurements must never take the headline line down (all ranks raise alike)
            import traceback
            traceback.print_exc()
            extras = {"error": repr(exc)[:300]}
        if world == 1 and "error" not in extras:
            extras["k1_batch_sweep"] = batch_sweep(dev, fence, max(args.extra_steps, 20))
            extras["roofline_per_config"] = config_rooflines(dev, fence, max(args.extra_steps, 40))
            # BASELINE configs[4]: 22.05 kHz stereo, n_fft 2048, 128 mel - banded fp32 (default) vs fp16 MFMA variant
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import gpu_c5
            extras["c5_stereo_2048_128mel"] = {"fp32_banded_default": gpu_c5.run("fp32", 40), "fp16_mfma": gpu_c5.run("fp16_mfma", 40),
                                               "default": "fp32 banded (meets north_star's 1e-5; the fp16-MFMA variant states 2e-3 and is opt-in)"}
            # three plans on three HIP streams, round-robin: independent batches overlap.  Throughput only - kernel
            # durations read under overlap include queueing, so the headline and the roofline stay single-stream.
            extras["stream_pipeline"] = two_stream(dev, wavs, outs, fence, max(300, args.steps))
            # the step as a replayed hipGraph (one graph per rotating batch): the host launch path taken out
            extras["graph_replay"] = graph_replay(dev, plan, wavs, outs, fence, max(200, args.steps))
            # the round-1 configuration (one batch replayed, Infinity-Cache resident) beside the rotating one
            fence()
            t0 = time.perf_counter()
            for _ in range(100):
                plan.wav_to_logmel(wavs[0], minmax=True, log=True, out=outs[0])
            fence()
            dt = (time.perf_counter() - t0) / 100
            plan.timing_enable(1)
            for _ in range(54):
                plan.wav_to_logmel(wavs[0], minmax=True, log=True, out=outs[0])
            fence()
            kr = plan.timing_samples(0)
            plan.timing_enable(False)
            extras["c2_cache_resident_replay"] = {"k1_us": round(1e3 * float(kr.mean()), 2) if len(kr) else None,
                                                  "step_us": round(1e6 * dt, 2)}
    timer.cancel()
    if extras and not args.only_sweep:  # the c3 / c4 checker legs count like the c2 one: a failed leg is a failed run
        for key in ("c3_frontend_specaug_crnn_fwd", "c4_train_step"):
            leg = (extras.get(key) or {}).get("parity")
            if leg is not None:
                parity_ok.append(bool(leg.get("ok")))
        # (the legs of the opt-in split-bf16 extras are reported with the line but do not decide its exit code: extras never do)
    finish(extras)
    if coll:
        dist.destroy_process_group()
    if parity_ok and not all(parity_ok):  # a fast kernel whose results differ from the reference's is not done
        print("bench.py: a parity check against the oracle FAILED (see `parity`, `extra.c3_frontend_specaug_crnn_fwd.parity`, "
              "`extra.c4_train_step.parity`)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
