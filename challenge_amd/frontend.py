"""Torch-facing wrapper of the HIP frontend (C ABI: include/iris_frontend.h).

PyTorch is plumbing here: it owns device memory and streams; every numeric op on
the hot path is a hand-written HIP kernel in csrc/iris_frontend.hip.  Tensors
must be float32, contiguous and on a ROCm device -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as N

EPSILON = 1e-8


def _require_device_f32(x: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(x).__name__}")
    if not x.is_cuda:
        raise RuntimeError(
            f"{name} is on {x.device}: the feature frontend runs only as HIP kernels on a "
            "ROCm device (there is no CPU fallback); move the tensor to 'cuda'")
    if x.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {x.dtype}")
    return x.contiguous()


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _bands_arg(bands, batch: int, device: torch.device, name: str):
    """bands: None or int tensor/array [B, n, 2] of (offset, size)."""
    if bands is None:
        return None, C.c_void_p(0), 0
    t = torch.as_tensor(bands)
    if t.dim() != 3 or t.shape[0] != batch or t.shape[2] != 2:
        raise ValueError(f"{name} must have shape [batch={batch}, n, 2], got {tuple(t.shape)}")
    t = t.to(device=device, dtype=torch.int32).contiguous()
    if t.shape[1] == 0:
        return None, C.c_void_p(0), 0
    return t, C.c_void_p(t.data_ptr()), int(t.shape[1])


def mel_weight_matrix(num_mel_bins: int = 20, num_spectrogram_bins: int = 129,
                      sample_rate: float = 8000, lower_edge_hertz: float = 125.0,
                      upper_edge_hertz: float = 3800.0) -> np.ndarray:
    """W[F, M] as the reference's closure builds it (transforms.py:55-56), via
    the library's host routine (fp32 recipe).  Raises ValueError on bad edges."""
    out = np.empty((num_spectrogram_bins, num_mel_bins), np.float32)
    rc = N.lib().iris_mel_weight_matrix(int(num_mel_bins), int(num_spectrogram_bins),
                                        float(sample_rate), float(lower_edge_hertz),
                                        float(upper_edge_hertz),
                                        out.ctypes.data_as(C.POINTER(C.c_float)))
    N.check(rc, "iris_mel_weight_matrix")
    return out


_LIVE_PLANS: "weakref.WeakSet[FrontendPlan]" = weakref.WeakSet()


def check_plans(device=None) -> None:
    """Raise EpilogueTimeout if any live plan (on `device`, or anywhere) reports a failed fused-epilogue wait.
    Synchronises; meant for the places that synchronise anyway (`sj_train.fit` calls it once per epoch)."""
    dev = None if device is None else torch.device(device)
    for plan in list(_LIVE_PLANS):
        if plan._handle is None or plan.n_fft == 0:
            continue
        if dev is not None and dev.type == "cuda" and dev.index is not None and plan.device != dev:
            continue
        plan.raise_on_failure()


class FrontendPlan:
    """State of `Spectrogram(n_fft, power=None)` (data_utils.py:17) plus the
    `magphase_to_mel(...)` closure (transforms.py:51-56) on one device."""

    def __init__(self, n_fft: int = 512, hop: Optional[int] = None, n_mel: int = 80,
                 sample_rate: float = 16000, channels: int = 1, max_batch: int = 64,
                 max_len: int = 160000, device=None, lower_edge_hertz: float = 125.0,
                 upper_edge_hertz: float = 3800.0, mel_matrix: Optional[np.ndarray] = None):
        self._handle = None
        lib = N.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("FrontendPlan needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the feature frontend")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"FrontendPlan device must be a ROCm device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_fft, self.hop = int(n_fft), int(n_fft // 2 if hop is None else hop)
        self.n_mel, self.n_bins = int(n_mel), int(n_fft) // 2 + 1
        self.sample_rate, self.channels = float(sample_rate), int(channels)
        self.max_batch, self.max_len = int(max_batch), int(max_len)
        mel_ptr = None
        if mel_matrix is not None:
            mel_matrix = np.ascontiguousarray(mel_matrix, np.float32)
            if mel_matrix.shape != (self.n_bins, self.n_mel):
                raise ValueError(f"mel_matrix must be [{self.n_bins}, {self.n_mel}]")
            mel_ptr = mel_matrix.ctypes.data_as(C.POINTER(C.c_float))
        handle = C.c_void_p()
        rc = lib.iris_plan_create(C.byref(handle), self.device.index, self.n_fft, self.hop,
                                  self.n_mel, self.n_bins, self.sample_rate,
                                  float(lower_edge_hertz), float(upper_edge_hertz), self.channels,
                                  self.max_batch, self.max_len, mel_ptr)
        N.check(rc, "iris_plan_create")
        self._handle = handle
        self._lock = threading.Lock()
        self.mel_precision = "fp32"
        masked = "ROC_GLOBAL_CU_MASK" in os.environ or "HSA_CU_MASK" in os.environ  # the library starts such plans on two kernels
        env = os.environ.get("IRIS_EPILOGUE")
        self.epilogue = "two_kernels" if (env == "1" or (masked and env is None)) else ("in_place" if env == "2" else "fused")
        _LIVE_PLANS.add(self)

    @classmethod
    def mel_only(cls, n_mel: int, n_bins: int, channels: int, max_batch: int, device,
                 mel_matrix: np.ndarray) -> "FrontendPlan":
        """Plan for magphase_to_mel on spectra of an arbitrary bin count (no FFT ops)."""
        self = cls.__new__(cls)
        self._handle = None
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_fft, self.hop, self.n_mel, self.n_bins = 0, 1, int(n_mel), int(n_bins)
        self.sample_rate, self.channels = 0.0, int(channels)
        self.max_batch, self.max_len = int(max_batch), 1
        mel_matrix = np.ascontiguousarray(mel_matrix, np.float32)
        if mel_matrix.shape != (self.n_bins, self.n_mel):
            raise ValueError(f"mel_matrix must be [{self.n_bins}, {self.n_mel}]")
        handle = C.c_void_p()
        rc = N.lib().iris_plan_create(C.byref(handle), self.device.index, 0, 1, self.n_mel, self.n_bins, 1.0, 0.0,
                                      0.5, self.channels, self.max_batch, 1,
                                      mel_matrix.ctypes.data_as(C.POINTER(C.c_float)))
        N.check(rc, "iris_plan_create")
        self._handle = handle
        self._lock = threading.Lock()
        return self

    def set_mel_precision(self, precision: str = "fp32") -> None:
        """'fp32' (default): banded fp32 mel reduction, 1e-5.  'fp16_mfma': |X| and W in fp16 on the matrix cores
        (v_mfma_f32_16x16x32_f16, fp32 accumulate; BASELINE configs[4]), 2e-3; raises ValueError when the plan's
        shape / filterbank cannot use it.  Calls that carry SpecAugment bands always run fp32."""
        code = {"fp32": N.IRIS_MEL_F32, "fp16_mfma": N.IRIS_MEL_F16_MFMA}[precision]
        N.check(N.lib().iris_plan_set_mel_precision(self._handle, code), "iris_plan_set_mel_precision")
        self.mel_precision = precision

    def set_epilogue(self, mode: str = "fused") -> None:
        """'fused' (default): min-max / log inside the fused kernel, one launch per call.  'two_kernels': raw mel +
        per-wave partials, then the min-max / log kernel - for several plans running CONCURRENTLY on one device
        (see iris_plan_set_epilogue in include/iris_frontend.h); calls under hipGraph capture take it by themselves."""
        code = {"fused": N.IRIS_EPILOGUE_FUSED, "two_kernels": N.IRIS_EPILOGUE_TWO_KERNELS, "in_place": N.IRIS_EPILOGUE_IN_PLACE}[mode]
        N.check(N.lib().iris_plan_set_epilogue(self._handle, code), "iris_plan_set_epilogue")
        self.epilogue = mode

    def last_epilogue(self) -> Optional[str]:
        """Form the last `wav_to_logmel` call took: 'fused' (min-max / log from the chunk's LDS tile), 'in_place' (one launch,
        the workgroup finishes its rows of `out` in place: chunks too large for the tile), 'two_kernels'; None before the
        first call or when neither min-max nor log was applied."""
        form = C.c_int(-1)
        N.check(N.lib().iris_plan_last_epilogue(self._handle, C.byref(form)), "iris_plan_last_epilogue")
        return {N.IRIS_EPILOGUE_FUSED: "fused", N.IRIS_EPILOGUE_TWO_KERNELS: "two_kernels", N.IRIS_EPILOGUE_IN_PLACE: "in_place"}.get(form.value)

    def status(self) -> int:
        """0 = every bounded in-kernel wait of the fused epilogue completed so far; 1 = one gave up (clips written as
        NaN).  Synchronises with the device and resets the word; after a 1 the plan stays on the two-kernel form."""
        st = C.c_int(0)
        N.check(N.lib().iris_plan_status(self._handle, C.byref(st)), "iris_plan_status")
        if st.value:
            self.epilogue = "two_kernels"
        return st.value

    def raise_on_failure(self) -> None:
        """`status()` as an exception: EpilogueTimeout naming the plan.  Call where a synchronisation happens anyway
        (end of an epoch, after reading a loss); the hot path itself reports a failed EARLIER launch without any
        synchronisation - the next `wav_to_logmel` on the plan raises the same exception."""
        if self._handle is not None and self.status():
            raise N.EpilogueTimeout(
                f"FrontendPlan(n_fft={self.n_fft}, hop={self.hop}, n_mel={self.n_mel}, channels={self.channels}, "
                f"max_batch={self.max_batch}, device={self.device}): a fused min-max / log epilogue gave up waiting for "
                "its clip's other workgroups (not co-resident: concurrent kernels, a CU mask or another process on the "
                "device) and wrote NaN features; the plan now uses the two-kernel form")

    def set_epilogue_timeout(self, microseconds: int) -> None:
        """Bound of the fused epilogue's in-kernel waits (default 2 s).  0 gives up after the first sweep: the test hook
        that makes the failure path reachable on a healthy device."""
        N.check(N.lib().iris_plan_set_epilogue_timeout(self._handle, int(microseconds)), "iris_plan_set_epilogue_timeout")

    def close(self) -> None:
        if self._handle is not None:
            N.lib().iris_plan_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- introspection ---------------------------------------------------
    def num_frames(self, length: int) -> int:
        return 1 + int(length) // self.hop

    @property
    def mel_matrix(self) -> np.ndarray:
        out = np.empty((self.n_bins, self.n_mel), np.float32)
        N.check(N.lib().iris_plan_get_mel(self._handle, out.ctypes.data_as(C.POINTER(C.c_float))),
                "iris_plan_get_mel")
        return out

    def _check_wav(self, wav: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
        wav = _require_device_f32(wav, "wav")
        if wav.device != self.device:
            raise RuntimeError(f"wav is on {wav.device}, plan is on {self.device}")
        if wav.dim() != 3 or wav.shape[1] != self.channels:
            raise ValueError(f"wav must be [B, C={self.channels}, L], got {tuple(wav.shape)}")
        return wav, int(wav.shape[0]), int(wav.shape[2])

    # ---- ops ---------------------------------------------------------------
    def stft(self, wav: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """wav [B,C,L] -> spec [B,F,T,2C] (load_wav, data_utils.py:17-27).  `normalize`: the wav / (10 rms) of
        data_utils.py:22-23 folded in (per clip, all channels jointly; the spectrum is scaled as it is written)."""
        wav, b, length = self._check_wav(wav)
        t = self.num_frames(length)
        spec = torch.empty((b, self.n_bins, t, 2 * self.channels), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            rc = N.lib().iris_stft(self._handle, wav.data_ptr(), spec.data_ptr(), b, length,
                                   N.IRIS_F_NORMALIZE if normalize else 0, _stream_ptr(self.device))
        N.check(rc, "iris_stft")
        return spec

    def magmel(self, spec: torch.Tensor, is_magphase: bool = False, t_bands=None, f_bands=None) -> torch.Tensor:
        """spec [B,F,T,2C] -> mel [B,M,T,C] (complex_to_magphase + magphase_to_mel)."""
        spec = _require_device_f32(spec, "spec")
        if spec.dim() != 4 or spec.shape[1] != self.n_bins or spec.shape[3] != 2 * self.channels:
            raise ValueError(f"spec must be [B, {self.n_bins}, T, {2 * self.channels}], got {tuple(spec.shape)}")
        b, t = int(spec.shape[0]), int(spec.shape[2])
        mel = torch.empty((b, self.n_mel, t, self.channels), dtype=torch.float32, device=spec.device)
        tb, tbp, ntb = _bands_arg(t_bands, b, spec.device, "t_bands")
        fb, fbp, nfb = _bands_arg(f_bands, b, spec.device, "f_bands")
        if b == 0 or t == 0:
            return mel
        with self._lock, torch.cuda.device(self.device):
            rc = N.lib().iris_magmel(self._handle, spec.data_ptr(), mel.data_ptr(), b, t,
                                     1 if is_magphase else 0, tbp, ntb, fbp, nfb, _stream_ptr(self.device))
        N.check(rc, "iris_magmel")
        return mel

    def wav_to_logmel(self, wav: torch.Tensor, minmax: bool = True, log: bool = True,
                      normalize: bool = False, t_bands=None, f_bands=None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The fused hot path: wav [B,C,L] -> log-mel [B,M,T,C]."""
        wav, b, length = self._check_wav(wav)
        t = self.num_frames(length)
        if out is None:
            out = torch.empty((b, self.n_mel, t, self.channels), dtype=torch.float32, device=self.device)
        else:
            out = _require_device_f32(out, "out")
            if tuple(out.shape) != (b, self.n_mel, t, self.channels):
                raise ValueError(f"out must be {(b, self.n_mel, t, self.channels)}, got {tuple(out.shape)}")
        tb, tbp, ntb = _bands_arg(t_bands, b, self.device, "t_bands")
        fb, fbp, nfb = _bands_arg(f_bands, b, self.device, "f_bands")
        flags = (N.IRIS_F_MINMAX if minmax else 0) | (N.IRIS_F_LOG if log else 0) | \
            (N.IRIS_F_NORMALIZE if normalize else 0)
        with self._lock, torch.cuda.device(self.device):
            rc = N.lib().iris_wav_to_logmel(self._handle, wav.data_ptr(), out.data_ptr(), b, length, flags,
                                            tbp, ntb, fbp, nfb, _stream_ptr(self.device))
        if rc == N.IRIS_E_EPILOGUE_TIMEOUT:
            self.epilogue = "two_kernels"  # the library has switched the plan for good
        N.check(rc, "iris_wav_to_logmel")
        return out

    # ---- prepared launches ------------------------------------------------------
    def prepare(self, wav: torch.Tensor, out: Optional[torch.Tensor] = None, minmax: bool = True, log: bool = True,
                normalize: bool = False, t_bands=None, f_bands=None) -> "PreparedCall":
        """Validate once, launch many times: returns an object whose `.launch()` is nothing but the C-ABI call of
        `wav_to_logmel(wav, out=out, ...)` with pre-converted arguments (about 3 us of host time instead of ~12 us for the
        checked path) - for steady-state loops over long-lived buffers (serving, benchmarks).  Tensors are bound BY
        ADDRESS (refill them in place); launches go to the stream that is current NOW; not thread-safe."""
        return PreparedCall(self, wav, out, minmax, log, normalize, t_bands, f_bands)

    # ---- hipGraph ------------------------------------------------------------
    def capture(self, wav: torch.Tensor, out: Optional[torch.Tensor] = None, **kwargs) -> "CapturedStep":
        """Capture `wav_to_logmel(wav, out=out, **kwargs)` into a hipGraph (torch.cuda.CUDAGraph) once and return an
        object whose `.replay()` re-runs it on the current stream with no per-launch host work; `.out` is the output
        tensor.  The entry points never allocate or synchronise and keep no per-call host state, so a replay is
        the same device work.  `wav`, `out` and any bands tensors are captured BY ADDRESS: refill them in place."""
        return CapturedStep(self, wav, out, kwargs)

    # ---- bench hooks ---------------------------------------------------------
    def fused_kernel_name(self, with_bands: bool = False) -> str:
        buf = C.create_string_buffer(128)
        N.check(N.lib().iris_plan_kernel_name(self._handle, 1 if with_bands else 0, buf, 128), "iris_plan_kernel_name")
        return buf.value.decode()

    def timing_enable(self, enable=True) -> None:
        """True / 1: every launch of the main kernel carries an event pair; n > 1: every n-th; False / 0: off."""
        N.check(N.lib().iris_timing_enable(self._handle, int(enable)), "iris_timing_enable")

    def timing_read(self) -> Tuple[int, float]:
        n, ms = C.c_int(0), C.c_float(0)
        N.check(N.lib().iris_timing_read(self._handle, C.byref(n), C.byref(ms)), "iris_timing_read")
        return n.value, ms.value

    def timing_samples(self, kernel: int = 0) -> np.ndarray:
        """Durations (ms) of the sampled launches of kernel 0 (fused) / 1 (min-max + log) since timing_enable."""
        cap = 4096
        buf = np.zeros(cap, np.float32)
        n = C.c_int(0)
        N.check(N.lib().iris_timing_samples(self._handle, int(kernel), buf.ctypes.data_as(C.POINTER(C.c_float)), cap,
                                            C.byref(n)), "iris_timing_samples")
        return buf[:min(n.value, cap)].copy()


class PreparedCall:
    """One fused frontend call with its arguments converted once (see FrontendPlan.prepare)."""
    __slots__ = ("out", "_fn", "_args", "_keep")

    def __init__(self, plan: FrontendPlan, wav, out, minmax, log, normalize, t_bands, f_bands):
        wav, b, length = plan._check_wav(wav)
        t = plan.num_frames(length)
        if out is None:
            out = torch.empty((b, plan.n_mel, t, plan.channels), dtype=torch.float32, device=plan.device)
        else:
            out = _require_device_f32(out, "out")
            if tuple(out.shape) != (b, plan.n_mel, t, plan.channels):
                raise ValueError(f"out must be {(b, plan.n_mel, t, plan.channels)}, got {tuple(out.shape)}")
        tb, tbp, ntb = _bands_arg(t_bands, b, plan.device, "t_bands")
        fb, fbp, nfb = _bands_arg(f_bands, b, plan.device, "f_bands")
        flags = (N.IRIS_F_MINMAX if minmax else 0) | (N.IRIS_F_LOG if log else 0) | (N.IRIS_F_NORMALIZE if normalize else 0)
        self.out, self._keep = out, (plan, wav, tb, fb)     # keep everything the raw pointers refer to alive
        self._fn = N.lib().iris_wav_to_logmel
        self._args = (plan._handle, C.c_void_p(wav.data_ptr()), C.c_void_p(out.data_ptr()), b, length, flags, tbp, ntb, fbp, nfb,
                      _stream_ptr(plan.device))

    def launch(self) -> torch.Tensor:
        rc = self._fn(*self._args)
        if rc:
            N.check(rc, "iris_wav_to_logmel")
        return self.out


class CapturedStep:
    """One fused frontend call as a replayable hipGraph (see FrontendPlan.capture)."""

    def __init__(self, plan: FrontendPlan, wav: torch.Tensor, out: Optional[torch.Tensor], kwargs: dict):
        self.plan, self.wav = plan, wav
        dev = plan.device
        # bands must be device tensors that outlive the graph (they are read at replay time)
        for k in ("t_bands", "f_bands"):
            if kwargs.get(k) is not None:
                kwargs[k] = torch.as_tensor(kwargs[k]).to(device=dev, dtype=torch.int32).contiguous()
        self.kwargs = kwargs
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up outside the capture: geometry cache, lazy module load
            self.out = plan.wav_to_logmel(wav, out=out, **kwargs)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            plan.wav_to_logmel(wav, out=self.out, **kwargs)

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.out


# ---------------------------------------------------------------------------
# plan-free ops (run on the tensor's device / current stream)
# ---------------------------------------------------------------------------
def bias_relu_(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """In place max(x + bias[c], 0) on a channels-last activation: x is [B, C, H, W] with channels_last strides (or any
    dense tensor whose innermost axis is the channel axis).  One HIP launch (iris_bias_relu)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        raise TypeError("bias_relu_: x must be a float32 tensor on a ROCm device (no CPU fallback)")
    nhwc = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
    if not nhwc and not (x.is_contiguous() and x.shape[-1] == bias.shape[0]):
        raise ValueError("bias_relu_: x must be channels_last [B, C, H, W], or contiguous with the channel axis innermost")
    if (nhwc and x.shape[1] != bias.shape[0]) or bias.dtype != torch.float32 or not bias.is_contiguous():
        raise ValueError("bias_relu_: bias must be a contiguous float32 vector with one entry per channel")
    c = int(bias.shape[0])
    with torch.cuda.device(x.device):
        rc = N.lib().iris_bias_relu(x.data_ptr(), bias.data_ptr(), x.numel() // c, c, _stream_ptr(x.device))
    N.check(rc, "iris_bias_relu")
    return x


def bias_relu_nchw_(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """In place max(x + bias[c], 0) on a CONTIGUOUS [B, C, H, W] activation (iris_bias_relu_nchw)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("bias_relu_nchw_: x must be a contiguous float32 [B, C, H, W] tensor on a ROCm device")
    b, c, h, w = (int(v) for v in x.shape)
    with torch.cuda.device(x.device):
        rc = N.lib().iris_bias_relu_nchw(x.data_ptr(), bias.data_ptr(), b, c, h * w, _stream_ptr(x.device))
    N.check(rc, "iris_bias_relu_nchw")
    return x


def bias_relu_maxpool_nchw(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """maxpool2x2('same')(relu(x + bias)) reading a CONTIGUOUS [B, C, H, W] activation and returning the pooled tensor in
    channels_last memory format (iris_bias_relu_maxpool_nchw): the hand-over from an NCHW block to the NHWC rest."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("bias_relu_maxpool_nchw: x must be a contiguous float32 [B, C, H, W] tensor on a ROCm device")
    b, c, h, w = (int(v) for v in x.shape)
    y = torch.empty((b, c, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = N.lib().iris_bias_relu_maxpool_nchw(x.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w, c, _stream_ptr(x.device))
    N.check(rc, "iris_bias_relu_maxpool_nchw")
    return y


def bias_relu_maxpool(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """maxpool2x2('same')(relu(x + bias)) for a channels_last [B, C, H, W] activation in ONE pass (iris_bias_relu_maxpool):
    reads x once, writes a quarter of it.  Returns a channels_last [B, C, ceil(H/2), ceil(W/2)] tensor."""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or not x.is_cuda or x.dtype != torch.float32:
        raise ValueError("bias_relu_maxpool: x must be a float32 channels_last [B, C, H, W] device tensor")
    b, c, h, w = (int(v) for v in x.shape)
    y = torch.empty((b, c, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = N.lib().iris_bias_relu_maxpool(x.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w, c, _stream_ptr(x.device))
    N.check(rc, "iris_bias_relu_maxpool")
    return y


def conv3x3_small_bias_relu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, channels_last: bool = False) -> torch.Tensor:
    """relu(conv2d(x, weight, padding=1) + bias) for a contiguous [B, 1 or 2, H, W] input in ONE pass: the CRNN's first layer,
    which writes 16-32x what it reads.  Output contiguous (iris_conv3x3_small_bias_relu_nchw; W a multiple of 4) or
    channels_last (iris_conv3x3_small_bias_relu_nhwc; W <= 2048, output channels 4 .. 256 dividing 1024)."""
    if (x.dim() != 4 or not x.is_contiguous() or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] not in (1, 2)
            or tuple(weight.shape[1:]) != (x.shape[1], 3, 3) or not weight.is_contiguous()):
        raise ValueError("conv3x3_small_bias_relu: x must be a contiguous float32 device tensor [B, 1|2, H, W], "
                         "weight [Cout, Cin, 3, 3] contiguous")
    b, cin, h, w = (int(v) for v in x.shape)
    cout = int(weight.shape[0])
    if channels_last:
        if cout % 4 or cout > 256 or 1024 % cout or w > 2048:
            raise ValueError("conv3x3_small_bias_relu(channels_last): output channels 4 .. 256 dividing 1024, W <= 2048")
        y = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        fn, name = N.lib().iris_conv3x3_small_bias_relu_nhwc, "iris_conv3x3_small_bias_relu_nhwc"
    else:
        if w % 4:
            raise ValueError("conv3x3_small_bias_relu: W must be a multiple of 4 for the contiguous output")
        y = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device)
        fn, name = N.lib().iris_conv3x3_small_bias_relu_nchw, "iris_conv3x3_small_bias_relu_nchw"
    with torch.cuda.device(x.device):
        rc = fn(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), b, cin, cout, h, w, _stream_ptr(x.device))
    N.check(rc, name)
    return y


def conv3x3_small_bias_relu_nchw(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return conv3x3_small_bias_relu(x, weight, bias, channels_last=False)


def conv3x3_c32_bias_relu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, pool: bool = False,
                          out_chunked: bool = False) -> torch.Tensor:
    """relu(conv2d(x, weight, padding=1) + bias), with `pool` max-pooled 2x2 'same' behind it, for a channels_last
    [B, 32, H, W] input and a [32, 32, 3, 3] weight, on the fp32 matrix cores (iris_conv3x3_c32_bias_relu).  Returns a
    channels_last tensor, or with `out_chunked` the channel-chunked [B, 4, Ho, Wo, 8] activation of the Winograd layers."""
    if (x.dim() != 4 or x.shape[1] != 32 or not x.is_contiguous(memory_format=torch.channels_last) or not x.is_cuda
            or x.dtype != torch.float32 or tuple(weight.shape) != (32, 32, 3, 3) or not weight.is_contiguous()):
        raise ValueError("conv3x3_c32_bias_relu: x must be a float32 channels_last device tensor [B, 32, H, W], weight a contiguous "
                         "[32, 32, 3, 3]")
    b, _, h, w = (int(v) for v in x.shape)
    oh, ow = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    if out_chunked:  # [B, 4, Ho, Wo, 8]: what the Winograd layers behind it read (no conversion pass)
        y = torch.empty((b, 4, oh, ow, 8), dtype=torch.float32, device=x.device)
    else:
        y = torch.empty((b, 32, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = N.lib().iris_conv3x3_c32_bias_relu(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w,
                                                1 if pool else 0, 1 if out_chunked else 0, _stream_ptr(x.device))
    N.check(rc, "iris_conv3x3_c32_bias_relu")
    return y


def conv3x3_c32(x: torch.Tensor, weight: torch.Tensor, transposed: bool = False, bn_sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The bare conv2d(x, weight, padding=1) of a 32 -> 32 layer on the fp32 matrix cores (iris_conv3x3_c32: the inference
    kernel without bias / ReLU) for a channels_last [B, 32, H, W] input and a [32, 32, 3, 3] weight in any dense layout;
    `transposed`: the backward-data pass, conv2d(x, weight.flip(2, 3).transpose(0, 1), padding=1) with x = dz."""
    if (x.dim() != 4 or x.shape[1] != 32 or not x.is_contiguous(memory_format=torch.channels_last) or not x.is_cuda
            or x.dtype != torch.float32 or tuple(weight.shape) != (32, 32, 3, 3) or weight.dtype != torch.float32):
        raise ValueError("conv3x3_c32: x must be a float32 channels_last device tensor [B, 32, H, W], weight a float32 [32, 32, 3, 3]")
    b, _, h, w = (int(v) for v in x.shape)
    y = torch.empty((b, 32, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    so, si, sh, sw = (int(v) for v in weight.stride())
    with torch.cuda.device(x.device):
        if bn_sums is not None:   # + the statistics of the BatchNorm behind it into the zeroed [iris_bn_sums_len(32)] doubles
            if transposed or bn_sums.dtype != torch.float64 or bn_sums.numel() < int(N.lib().iris_bn_sums_len(32)):
                raise ValueError("conv3x3_c32: bn_sums must be float64 [iris_bn_sums_len(32)] (forward pass only)")
            rc = N.lib().iris_conv3x3_c32_bn(x.data_ptr(), weight.data_ptr(), so, si, sh, sw, y.data_ptr(), b, h, w, bn_sums.data_ptr(),
                                             _stream_ptr(x.device))
        else:
            rc = N.lib().iris_conv3x3_c32(x.data_ptr(), weight.data_ptr(), so, si, sh, sw, 1 if transposed else 0, y.data_ptr(), b, h, w,
                                          _stream_ptr(x.device))
    N.check(rc, "iris_conv3x3_c32")
    return y


def wino_pack_weights(weight: torch.Tensor) -> torch.Tensor:
    """U = G g G^T of a [Cout, Cin, 3, 3] convolution weight in the Winograd kernel's LDS order (iris_wino_pack_weights, on the
    host, once per layer); returns a device tensor of 16 Cin Cout floats on the weight's device."""
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    if tuple(weight.shape[2:]) != (3, 3):
        raise ValueError("wino_pack_weights: a [Cout, Cin, 3, 3] weight is expected")
    host = np.ascontiguousarray(weight.detach().to(torch.float32).cpu().contiguous().numpy())
    out = np.empty(int(N.lib().iris_wino_packed_len(cin, cout)), np.float32)
    N.check(N.lib().iris_wino_pack_weights(host.ctypes.data, cin, cout, out.ctypes.data), "iris_wino_pack_weights")
    return torch.from_numpy(out).to(weight.device)


def to_chunked(x: torch.Tensor) -> torch.Tensor:
    """channels_last [B, C, H, W] (or any layout) -> the channel-chunked activation [B, C / 8, H, W, 8] of the Winograd layers."""
    b, c, h, w = (int(v) for v in x.shape)
    return x.permute(0, 2, 3, 1).reshape(b, h, w, c // 8, 8).permute(0, 3, 1, 2, 4).contiguous()


def wino_pack_weights_device(weight: torch.Tensor, transposed: bool = False, out: Optional[torch.Tensor] = None,
                             split_bf16: bool = False) -> torch.Tensor:
    """The same packing on the device, on the current stream, from the weight as it lies in memory (any strides: a
    channels_last parameter needs no copy) - what a training step does every step.  `transposed`: the weights of the
    backward-data pass (dx = conv(dz, W'), W'[ci][co][i][j] = W[co][ci][2 - i][2 - j]).  `split_bf16`: U split into three bf16
    terms in the operand order of iris_conv3x3_wino_b3 (the convolution on the BF16 matrix cores at fp32 accuracy)."""
    if not (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
        raise ValueError("wino_pack_weights_device: a float32 device weight [Cout, Cin, 3, 3] is expected (no CPU fallback)")
    co, ci = int(weight.shape[0]), int(weight.shape[1])
    cin, cout = (co, ci) if transposed else (ci, co)
    lib = N.lib()
    n = int((lib.iris_wino_b3_packed_len if split_bf16 else lib.iris_wino_packed_len)(cin, cout))
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=weight.device)
    so, si, sh, sw = (int(v) for v in weight.stride())
    pack = lib.iris_wino_b3_pack_weights_device if split_bf16 else lib.iris_wino_pack_weights_device
    with torch.cuda.device(weight.device):
        rc = pack(weight.data_ptr(), so, si, sh, sw, cin, cout, 1 if transposed else 0, out.data_ptr(), _stream_ptr(weight.device))
    N.check(rc, "iris_wino_b3_pack_weights_device" if split_bf16 else "iris_wino_pack_weights_device")
    return out


def wino_packed_len(cin: int, cout: int, split_bf16: bool = False) -> int:
    """floats of the packed weights of a cin -> cout layer (iris_wino_packed_len / iris_wino_b3_packed_len)."""
    lib = N.lib()
    return int((lib.iris_wino_b3_packed_len if split_bf16 else lib.iris_wino_packed_len)(int(cin), int(cout)))


def wino_pack_weights_device_multi(jobs, split_bf16: bool = False) -> None:
    """`jobs`: [(weight [Cout, Cin, 3, 3] float32 device tensor, transposed, out float32 device tensor of wino_packed_len floats)]:
    every packing in ONE launch on the current stream (iris_wino_pack_weights_device_multi), same results as
    `wino_pack_weights_device` per job."""
    if not jobs:
        return
    arr = (N.PackJob * len(jobs))()
    dev = jobs[0][0].device
    for k, (weight, transposed, out) in enumerate(jobs):
        if not (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
                and out.is_cuda and out.dtype == torch.float32 and weight.device == dev and out.device == dev):
            raise ValueError("wino_pack_weights_device_multi: float32 device tensors on one device are expected (no CPU fallback)")
        co, ci = int(weight.shape[0]), int(weight.shape[1])
        cin, cout = (co, ci) if transposed else (ci, co)
        if out.numel() < wino_packed_len(cin, cout, split_bf16):
            raise ValueError("wino_pack_weights_device_multi: output buffer too small")
        so, si, sh, sw = (int(v) for v in weight.stride())
        arr[k] = N.PackJob(weight.data_ptr(), out.data_ptr(), so, si, sh, sw, cin, cout, 1 if transposed else 0, 0)
    with torch.cuda.device(dev):
        rc = N.lib().iris_wino_pack_weights_device_multi(C.cast(arr, C.c_void_p), len(jobs), 1 if split_bf16 else 0, _stream_ptr(dev))
    N.check(rc, "iris_wino_pack_weights_device_multi")


def conv3x3_wino(x: torch.Tensor, packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, pool: bool = False,
                 out_nhwc: bool = False, relu: bool = True, split_bf16: bool = False, bn_sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv2d(x, weight, padding=1) (+ bias, + ReLU, + MaxPool 2x2 'same') as Winograd F(2x2, 3x3) on the fp32 matrix cores
    (iris_conv3x3_wino).  x: channel-chunked [B, Cin / 8, H, W, 8], or a channels_last [B, Cin, H, W] tensor (read where it
    lies: IRIS_WINO_IN_NHWC); packed: `wino_pack_weights[_device]`; returns the chunked [B, cout / 8, Ho, Wo, 8] or, with
    `out_nhwc`, a channels_last [B, cout, Ho, Wo] tensor."""
    if not (x.is_cuda and x.dtype == torch.float32):
        raise ValueError("conv3x3_wino: x must be a float32 device tensor (no CPU fallback)")
    if x.dim() == 5 and x.shape[4] == 8 and x.is_contiguous():
        b, cbk, h, w, _ = (int(v) for v in x.shape)
        cin, in_nhwc = 8 * cbk, False
    elif x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        b, cin, h, w = (int(v) for v in x.shape)
        in_nhwc = True
    else:
        raise ValueError("conv3x3_wino: x must be contiguous [B, Cin / 8, H, W, 8] or channels_last [B, Cin, H, W]")
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    if out_nhwc:
        y = torch.empty((b, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    else:
        y = torch.empty((b, cout // 8, ho, wo, 8), dtype=torch.float32, device=x.device)
    flags = (N.IRIS_WINO_POOL if pool else 0) | (N.IRIS_WINO_OUT_NHWC if out_nhwc else 0) | \
        (N.IRIS_WINO_IN_NHWC if in_nhwc else 0) | (N.IRIS_WINO_RELU if relu else 0)
    lib = N.lib()
    with torch.cuda.device(x.device):
        if bn_sums is not None:   # the bare convolution + the statistics of the BatchNorm behind it (zeroed float64 [iris_bn_sums_len(cout)])
            if bias is not None or bn_sums.dtype != torch.float64 or bn_sums.numel() < int(lib.iris_bn_sums_len(int(cout))):
                raise ValueError("conv3x3_wino: bn_sums goes with the bare convolution (no bias) and must be float64 [iris_bn_sums_len(cout)]")
            fn = lib.iris_conv3x3_wino_b3_bn if split_bf16 else lib.iris_conv3x3_wino_bn
            rc = fn(x.data_ptr(), packed.data_ptr(), y.data_ptr(), b, h, w, cin, int(cout), flags, bn_sums.data_ptr(), _stream_ptr(x.device))
        else:
            fn = lib.iris_conv3x3_wino_b3 if split_bf16 else lib.iris_conv3x3_wino   # `packed` must come from the matching packer
            rc = fn(x.data_ptr(), packed.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                    b, h, w, cin, int(cout), flags, _stream_ptr(x.device))
    N.check(rc, "iris_conv3x3_wino_b3" if split_bf16 else "iris_conv3x3_wino")
    return y


# (device index, stream) -> scratch tensors of conv3x3_wino_wrw.  Grow-only: a captured hipGraph keeps its pointers.  One set per
# STREAM: the partial sums of two backward passes running on different streams of a device (a graph replay beside an eager
# step, two models on side streams) must not share a buffer; calls on one stream are ordered by the stream.
_WRW_WORKSPACES = {}


def _wrw_workspace(device: torch.device, floats: int) -> torch.Tensor:
    held = _WRW_WORKSPACES.setdefault((device.index, int(torch.cuda.current_stream(device).cuda_stream)), [])
    if not held or held[-1].numel() < floats:
        held.append(torch.empty(max(floats, 1 << 24), dtype=torch.float32, device=device))
    return held[-1]


def conv3x3_wino_wrw(x: torch.Tensor, dy: torch.Tensor, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The weight gradient of z = conv2d(x, weight, padding=1) given dy = dL/dz, as Winograd F(2x2, 3x3) on the fp32 matrix
    cores (iris_conv3x3_wino_wrw).  x [B, Cin, H, W] and dy [B, Cout, H, W]: channels_last float32 device tensors, Cin and
    Cout multiples of 32.  Returns dW [Cout, Cin, 3, 3] with the strides of `like` (the weight) or channels_last.
    The partial sums go through one scratch buffer per (device, stream): calls on one stream are ordered by it, calls on
    different streams do not share a buffer."""
    if not (x.is_cuda and x.dtype == torch.float32 and dy.dtype == torch.float32 and dy.device == x.device):
        raise ValueError("conv3x3_wino_wrw: x and dy must be float32 tensors on one device (no CPU fallback)")
    if not (x.dim() == 4 and dy.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and dy.is_contiguous(memory_format=torch.channels_last) and x.shape[0] == dy.shape[0] and x.shape[2:] == dy.shape[2:]):
        raise ValueError("conv3x3_wino_wrw: x [B, Cin, H, W] and dy [B, Cout, H, W] must be channels_last and of one geometry")
    b, cin, h, w = (int(v) for v in x.shape)
    cout = int(dy.shape[1])
    if like is not None and tuple(like.shape) == (cout, cin, 3, 3) and like.dtype == torch.float32:
        dw = torch.empty_like(like)   # preserves the parameter's strides: AccumulateGrad takes it without a copy
    else:
        dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
    lib = N.lib()
    with torch.cuda.device(x.device):
        ws = _wrw_workspace(x.device, int(lib.iris_wino_wrw_workspace_len(b, h, w, cin, cout)))
        so, si, sh, sw = (int(v) for v in dw.stride())
        rc = lib.iris_conv3x3_wino_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), so, si, sh, sw, b, h, w, cin, cout, 0,
                                       ws.data_ptr(), ws.numel(), _stream_ptr(x.device))
    N.check(rc, "iris_conv3x3_wino_wrw")
    return dw


def conv3x3_wino_bias_relu(x: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, cout: int, pool: bool = False,
                           out_nhwc: bool = False, split_bf16: bool = False) -> torch.Tensor:
    """The inference form: relu(conv2d(x, weight, padding=1) + bias), with `pool` max-pooled, on the chunked layout."""
    if not (x.dim() == 5 and x.shape[-1] == 8):
        raise ValueError("conv3x3_wino_bias_relu: x must be the channel-chunked activation [B, Cin / 8, H, W, 8]")
    return conv3x3_wino(x, packed, bias, cout, pool=pool, out_nhwc=out_nhwc, relu=True, split_bf16=split_bf16)


def _check_bilstm(gx, w_hh, who):
    if (gx.dim() != 4 or tuple(gx.shape[2:]) != (2, 512) or tuple(w_hh.shape) != (2, 512, 128) or not gx.is_cuda
            or gx.dtype != torch.float32 or w_hh.dtype != torch.float32 or w_hh.device != gx.device):
        raise ValueError(f"{who}: gx must be a float32 device tensor [B, T, 2, 512], w_hh [2, 512, 128] beside it")


def bilstm128_forward(gx: torch.Tensor, w_hh: torch.Tensor, save: bool = False):
    """Recurrent half of Bidirectional(LSTM(128, return_sequences=True)) in ONE launch (iris_bilstm128_forward).
    gx [B, T, 2, 512]: input pre-activations x_t W_ih^T + b_ih + b_hh per direction, gate rows i, f, g, o;
    w_hh [2, 512, 128]: recurrent matrices.  Returns [B, T, 256] = (h forward, h backward) per step, h_0 = c_0 = 0;
    with `save` also the activations [B, T, 2, 5, 128] = (i, f, g, o, c) that `bilstm128_backward` needs."""
    _check_bilstm(gx, w_hh, "bilstm128_forward")
    gx, w_hh = gx.contiguous(), w_hh.contiguous()
    b, t = int(gx.shape[0]), int(gx.shape[1])
    out = torch.empty((b, t, 256), dtype=torch.float32, device=gx.device)
    act = torch.empty((b, t, 2, 5, 128), dtype=torch.float32, device=gx.device) if save else None
    with torch.cuda.device(gx.device):
        rc = N.lib().iris_bilstm128_forward(gx.data_ptr(), w_hh.data_ptr(), out.data_ptr(),
                                            act.data_ptr() if save else None, b, t, _stream_ptr(gx.device))
    N.check(rc, "iris_bilstm128_forward")
    return (out, act) if save else out


def bilstm128_backward(dout: torch.Tensor, act: torch.Tensor, w_hh: torch.Tensor) -> torch.Tensor:
    """Back-propagation through time of `bilstm128_forward` in ONE launch: dout [B, T, 256] and the saved activations ->
    the gradient of the input pre-activations, dgx [B, T, 2, 512]."""
    b, t = int(dout.shape[0]), int(dout.shape[1])
    if (tuple(dout.shape) != (b, t, 256) or tuple(act.shape) != (b, t, 2, 5, 128) or tuple(w_hh.shape) != (2, 512, 128)
            or not dout.is_cuda or dout.dtype != torch.float32 or act.dtype != torch.float32):
        raise ValueError("bilstm128_backward: dout [B, T, 256], act [B, T, 2, 5, 128], w_hh [2, 512, 128] (float32, device)")
    dout, act, w_hh = dout.contiguous(), act.contiguous(), w_hh.contiguous()
    dgx = torch.empty((b, t, 2, 512), dtype=torch.float32, device=dout.device)
    with torch.cuda.device(dout.device):
        rc = N.lib().iris_bilstm128_backward(dout.data_ptr(), act.data_ptr(), w_hh.data_ptr(), dgx.data_ptr(), b, t,
                                             _stream_ptr(dout.device))
    N.check(rc, "iris_bilstm128_backward")
    return dgx


class PipelinedFrontend:
    """Independent batches through the fused path on `n_streams` HIP streams, one `FrontendPlan` each (a plan's
    workspace belongs to one stream, include/iris_frontend.h), in the two-kernel form of the step: one stream's
    min-max/log kernel and launch gaps run in the shadow of another's fused kernel.

        pipe = PipelinedFrontend(2, n_fft=1024, hop=256, n_mel=64, channels=1, max_batch=32, max_len=160000, device=dev)
        for wav, out in batches:                 # device tensors; `out` is written asynchronously
            pipe.submit(wav, out=out)
        pipe.synchronize()                       # ... or order consumers after pipe.streams[i] with events

    The caller's current stream at `submit` time is waited for (inputs produced on it are ready), and every submit's
    stream is joined back by `synchronize()` / `join()` so that later work on the current stream sees the outputs."""

    def __init__(self, n_streams: int = 2, **plan_kwargs):
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        self.plans = [FrontendPlan(**plan_kwargs) for _ in range(n_streams)]
        if n_streams > 1:  # concurrent launches on one device: no in-kernel waits between workgroups
            for p in self.plans:
                p.set_epilogue("two_kernels")
        self.device = self.plans[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n_streams)]
        self._next = 0

    def submit(self, wav: torch.Tensor, wait_current: bool = True, **kwargs) -> torch.Tensor:
        """wait_current=False skips the event wait on the current stream and the allocator bookkeeping (about 7 us
        of host time per call, enough to make a 23 us step host-bound): for inputs and `out` buffers that are
        long-lived and already complete, as in a steady-state loop over preallocated batches."""
        i, self._next = self._next, (self._next + 1) % len(self.plans)
        stream = self.streams[i]
        if wait_current:
            stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(stream):
            out = self.plans[i].wav_to_logmel(wav, **kwargs)
        if wait_current:
            wav.record_stream(stream)
            out.record_stream(stream)
        return out

    def join(self) -> None:
        """Order the current stream after everything submitted so far (no host wait)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()


def resample(wav: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """torchaudio.compliance.kaldi.resample_waveform(wav, orig_freq, new_freq) (data_utils.py:20-21) for a device waveform
    [chan, samples] (or [samples]): the Hann-windowed-sinc polyphase filter of torchaudio.functional.resample
    (lowpass_filter_width 6, rolloff 0.99) as one HIP launch (iris_resample); returns [chan, ceil(new_freq samples / orig_freq)]."""
    wav = _require_device_f32(wav, "wav")
    if int(orig_freq) != orig_freq or int(new_freq) != new_freq or orig_freq <= 0 or new_freq <= 0:
        raise ValueError("resample: the sample rates must be positive integers")   # (torchaudio raises for non-integer rates too)
    flat = wav.dim() == 1
    if wav.dim() not in (1, 2) or wav.numel() == 0:
        raise ValueError("resample: wav must be a non-empty [chan, samples] or [samples] tensor")
    w2 = (wav.unsqueeze(0) if flat else wav).contiguous()
    chan, n = int(w2.shape[0]), int(w2.shape[1])
    lib = N.lib()
    n_out = int(lib.iris_resample_len(n, int(orig_freq), int(new_freq)))
    out = torch.empty((chan, n_out), dtype=torch.float32, device=wav.device)
    with torch.cuda.device(wav.device):
        rc = lib.iris_resample(w2.data_ptr(), chan, n, int(orig_freq), int(new_freq), out.data_ptr(), _stream_ptr(wav.device))
    N.check(rc, "iris_resample")
    return out[0] if flat else out


def normalize(wav: torch.Tensor) -> torch.Tensor:
    """wav / (10 rms) with the rms over the whole tensor for [C,L] input
    (data_utils.py:32-34) or per leading item for [B,C,L]."""
    wav = _require_device_f32(wav, "wav")
    if wav.dim() not in (2, 3):
        raise ValueError("wav must be [C, L] or [B, C, L]")
    n_rows = 1 if wav.dim() == 2 else int(wav.shape[0])
    row_len = wav.numel() // max(n_rows, 1)
    out = torch.empty_like(wav)
    lib = N.lib()
    ws = torch.empty(max(int(lib.iris_normalize_workspace(n_rows, row_len)), 1), dtype=torch.float32,
                     device=wav.device)
    with torch.cuda.device(wav.device):
        rc = lib.iris_normalize(wav.data_ptr(), out.data_ptr(), n_rows, row_len, ws.data_ptr(), ws.numel(),
                                _stream_ptr(wav.device))
    N.check(rc, "iris_normalize")
    return out


def minmax_log(x: torch.Tensor, do_minmax: bool = True, do_log: bool = True, eps_div: float = EPSILON,
               eps_log: float = EPSILON) -> torch.Tensor:
    """Out-of-place min-max over every axis except 0, then ln(x + eps)."""
    x = _require_device_f32(x, "x").clone()
    if x.dim() < 1 or x.numel() == 0:
        return x
    n_rows = int(x.shape[0])
    row_len = x.numel() // n_rows
    lib = N.lib()
    ws = torch.empty(max(int(lib.iris_minmax_log_workspace(n_rows, row_len)), 1), dtype=torch.float32,
                     device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.iris_minmax_log(x.data_ptr(), n_rows, row_len, 1 if do_minmax else 0, 1 if do_log else 0,
                                 float(eps_div), float(eps_log), ws.data_ptr(), ws.numel(),
                                 _stream_ptr(x.device))
    N.check(rc, "iris_minmax_log")
    return x


def complex_to_magphase(x: torch.Tensor) -> torch.Tensor:
    x = _require_device_f32(x, "complex_tensor")
    c2 = int(x.shape[-1])
    if c2 % 2:
        raise ValueError("last axis must hold [re block, im block] (even length)")
    out = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            rc = N.lib().iris_complex_to_magphase(x.data_ptr(), out.data_ptr(), x.numel() // c2, c2 // 2,
                                                  _stream_ptr(x.device))
        N.check(rc, "iris_complex_to_magphase")
    return out


def magphase_to_complex(x: torch.Tensor) -> torch.Tensor:
    x = _require_device_f32(x, "magphase")
    c2 = int(x.shape[-1])
    if c2 % 2:
        raise ValueError("last axis must hold [mag block, phase block] (even length)")
    out = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            rc = N.lib().iris_magphase_to_complex(x.data_ptr(), out.data_ptr(), x.numel() // c2, c2 // 2,
                                                  _stream_ptr(x.device))
        N.check(rc, "iris_magphase_to_complex")
    return out


def mask_apply(x: torch.Tensor, axis: int, bands, outer_per_group: Optional[int] = None) -> torch.Tensor:
    """Zero the bands [offset, offset+size) along `axis` (out of place).

    bands: int [n, 2] (one group for the whole tensor) or [G, n, 2] with
    `outer_per_group` leading-index rows per group."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError("mask_apply runs as a HIP kernel: x must be a tensor on a ROCm device")
    if x.element_size() not in (4, 8):
        raise TypeError(f"mask_apply supports 4- and 8-byte dtypes, got {x.dtype}")
    x = x.contiguous().clone()
    axis = axis % x.dim()
    n_outer = int(np.prod(x.shape[:axis], dtype=np.int64)) if axis > 0 else 1
    n_inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64)) if axis + 1 < x.dim() else 1
    b = torch.as_tensor(bands).to(device=x.device, dtype=torch.int32)
    if b.dim() == 2:
        b = b[None]
    if b.dim() != 3 or b.shape[-1] != 2:
        raise ValueError("bands must be [n, 2] or [G, n, 2]")
    b = b.contiguous()
    groups = int(b.shape[0])
    if outer_per_group is None:
        outer_per_group = max(n_outer // groups, 1) if groups > 1 else max(n_outer, 1)
    if groups * outer_per_group < n_outer:
        raise ValueError("bands groups do not cover the outer extent")
    if x.numel() and b.shape[1]:
        with torch.cuda.device(x.device):
            rc = N.lib().iris_mask_apply(x.data_ptr(), n_outer, int(x.shape[axis]), n_inner, x.element_size(),
                                         b.data_ptr(), int(b.shape[1]), int(outer_per_group),
                                         _stream_ptr(x.device))
        N.check(rc, "iris_mask_apply")
    return x


# ---------------------------------------------------------------------------
# plan cache for the closure-style API of transforms.py
# ---------------------------------------------------------------------------
_plans = {}
_plans_lock = threading.Lock()


def get_plan(device, n_fft: int, hop: Optional[int], n_mel: int, sample_rate: float, channels: int,
             batch: int, length: int, **mel_kw) -> FrontendPlan:
    """Cached plan with capacity >= (batch, length); grown geometrically."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    hop = n_fft // 2 if hop is None else hop
    key = (device.index, n_fft, hop, n_mel, float(sample_rate), channels, tuple(sorted(mel_kw.items())))
    with _plans_lock:
        plan = _plans.get(key)
        if plan is None or plan.max_batch < batch or plan.max_len < length:
            cap_b = max(batch, plan.max_batch if plan else 0)
            cap_l = max(length, plan.max_len if plan else 0, n_fft)
            if plan is not None:  # grow with headroom
                cap_b, cap_l = max(cap_b, 2 * plan.max_batch), max(cap_l, plan.max_len)
            plan = FrontendPlan(n_fft, hop, n_mel, sample_rate, channels, cap_b, cap_l, device, **mel_kw)
            _plans[key] = plan
        return plan
