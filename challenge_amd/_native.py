"""ctypes binding of the C ABI in include/iris_frontend.h.

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C
challenge_amd/csrc``).  Loading fails loudly: there is no pure-Python or CPU
fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# IRIS_LIB overrides the library (e.g. the diagnostic build libiris_frontend_diag.so)
LIB_PATH = os.environ.get("IRIS_LIB") or os.path.join(_HERE, "csrc", "libiris_frontend.so")

IRIS_F_MINMAX, IRIS_F_LOG, IRIS_F_NORMALIZE = 1, 2, 4
IRIS_MEL_F32, IRIS_MEL_F16_MFMA = 0, 1
IRIS_EPILOGUE_FUSED, IRIS_EPILOGUE_TWO_KERNELS, IRIS_EPILOGUE_IN_PLACE = 0, 1, 2
IRIS_WINO_POOL, IRIS_WINO_OUT_NHWC, IRIS_WINO_IN_NHWC, IRIS_WINO_RELU = 1, 2, 4, 8
IRIS_E_EPILOGUE_TIMEOUT = -5

# every symbol include/iris_frontend.h declares, with (restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class PackJob(C.Structure):
    """iris_pack_job of include/iris_frontend.h (iris_wino_pack_weights_device_multi)."""
    _fields_ = [("weight", C.c_void_p), ("packed", C.c_void_p), ("stride_o", C.c_long), ("stride_i", C.c_long), ("stride_h", C.c_long),
                ("stride_w", C.c_long), ("cin", C.c_int), ("cout", C.c_int), ("transposed", C.c_int), ("first_block", C.c_int)]
_fp = C.POINTER(C.c_float)
SIGNATURES = {
    "iris_abi_version": (_i, []),
    "iris_last_error": (C.c_char_p, []),
    "iris_mel_weight_matrix": (_i, [_i, _i, _f, _f, _f, _fp]),
    "iris_plan_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _i, _fp]),
    "iris_plan_destroy": (_i, [_vp]),
    "iris_plan_set_mel_precision": (_i, [_vp, _i]),
    "iris_plan_set_epilogue": (_i, [_vp, _i]),
    "iris_plan_status": (_i, [_vp, C.POINTER(_i)]),
    "iris_plan_set_epilogue_timeout": (_i, [_vp, C.c_ulonglong]),
    "iris_plan_get_mel": (_i, [_vp, _fp]),
    "iris_plan_num_frames": (_i, [_vp, _i]),
    "iris_normalize_workspace": (_sz, [_i, _sz]),
    "iris_normalize": (_i, [_vp, _vp, _i, _sz, _vp, _sz, _vp]),
    "iris_wino_pack_weights_device_multi": (_i, [_vp, _i, _i, _vp]),
    "iris_agc_clip_adam": (_i, [_vp, _sz, _f, _f, _f, _i, _vp, _f, C.c_double, C.c_double, _f, _vp, _vp]),
    "iris_resample_len": (C.c_longlong, [C.c_longlong, _i, _i]),
    "iris_resample": (_i, [_vp, _i, C.c_longlong, _i, _i, _vp, _vp]),
    "iris_stft": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "iris_complex_to_magphase": (_i, [_vp, _vp, _sz, _i, _vp]),
    "iris_magphase_to_complex": (_i, [_vp, _vp, _sz, _i, _vp]),
    "iris_magmel": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "iris_minmax_log_workspace": (_sz, [_i, _sz]),
    "iris_minmax_log": (_i, [_vp, _i, _sz, _i, _i, _f, _f, _vp, _sz, _vp]),
    "iris_wav_to_logmel": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "iris_mask_apply": (_i, [_vp, _sz, _sz, _sz, _i, _vp, _i, _sz, _vp]),
    "iris_agc_clip": (_i, [_vp, _sz, _f, _f, _f, _vp]),
    "iris_mix_frame_active": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "iris_mix_workspace": (_sz, [_i, _i]),
    "iris_mix_specs": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "iris_mix_wave_frame_active": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "iris_mix_waves": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "iris_bn_sums_len": (_sz, [_i]),
    "iris_bn_stats": (_i, [_vp, _sz, _i, _vp, _vp]),
    "iris_bn_relu_apply": (_i, [_vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_apply_sums0": (_i, [_vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_bwd_reduce": (_i, [_vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_bwd_dx": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_pool_apply": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_pool_apply_sums0": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_pool_bwd_reduce": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "iris_bn_relu_pool_bwd_dx": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "iris_conv3x3_c32_bias_relu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "iris_conv3x3_c32": (_i, [_vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _i, _vp, _i, _i, _i, _vp]),
    "iris_conv3x3_c32_bn": (_i, [_vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _vp, _i, _i, _i, _vp, _vp]),
    "iris_wino_packed_len": (_sz, [_i, _i]),
    "iris_wino_pack_weights": (_i, [_vp, _i, _i, _vp]),
    "iris_wino_pack_weights_device": (_i, [_vp, C.c_long, C.c_long, C.c_long, C.c_long, _i, _i, _i, _vp, _vp]),
    "iris_conv3x3_wino": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "iris_wino_b3_packed_len": (_sz, [_i, _i]),
    "iris_wino_b3_pack_weights_device": (_i, [_vp, C.c_long, C.c_long, C.c_long, C.c_long, _i, _i, _i, _vp, _vp]),
    "iris_conv3x3_wino_b3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "iris_conv3x3_wino_bn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "iris_conv3x3_wino_b3_bn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "iris_wino_wrw_workspace_len": (_sz, [_i, _i, _i, _i, _i]),
    "iris_conv3x3_wino_wrw": (_i, [_vp, _vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "iris_conv0_dweight_len": (_sz, [_i, _i]),
    "iris_conv0_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "iris_conv0_bn_relu": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "iris_conv0_bn_relu_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "iris_conv3x3_small_bias_relu_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "iris_conv3x3_small_bias_relu_nchw": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "iris_bilstm128_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "iris_bilstm128_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "iris_mix_draw": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, C.c_uint64, _vp, _vp, _vp, _vp]),
    "iris_augment_draw": (_i, [_i, _i, _i, _i, _i, _i, _i, C.c_uint64, _vp, _vp, _vp, _vp]),
    "iris_bias_relu": (_i, [_vp, _vp, _sz, _i, _vp]),
    "iris_bias_relu_maxpool": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "iris_bias_relu_nchw": (_i, [_vp, _vp, _sz, _i, _sz, _vp]),
    "iris_bias_relu_maxpool_nchw": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "iris_plan_kernel_name": (_i, [_vp, _i, C.c_char_p, _i]),
    "iris_plan_last_epilogue": (_i, [_vp, C.POINTER(_i)]),
    "iris_timing_enable": (_i, [_vp, _i]),
    "iris_timing_read": (_i, [_vp, C.POINTER(_i), _fp]),
    "iris_timing_samples": (_i, [_vp, _i, _fp, _i, C.POINTER(_i)]),
}

_lib = None
_lock = threading.Lock()


class IrisError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


class EpilogueTimeout(IrisError):
    """A fused-epilogue launch of a plan gave up a bounded wait (IRIS_E_EPILOGUE_TIMEOUT): its workgroups were not
    co-resident - concurrent kernels, a CU mask or another process on the device.  Features produced by that plan since
    its last clean status are suspect (the affected clips are NaN); the plan has switched to the two-kernel form."""


def lib() -> C.CDLL:
    """Load libiris_frontend.so (once).  Raises ImportError if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} is missing: the HIP frontend is not built "
                        "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
                        "`make -C challenge_amd/csrc`).  There is no CPU fallback.")
                # torch first: its wheel bundles its own libamdhip64.so.7 + HSA runtime.  If this
                # library were loaded before torch it would pull /opt/rocm's runtime in by
                # RUNPATH, torch would then be bound to that one by soname but to its bundled
                # HSA, and hipSetDevice fails.  One HIP runtime per process: torch's.
                import torch  # noqa: F401
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)  # AttributeError if the symbol is absent
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status == 0:
        return
    msg = lib().iris_last_error().decode("utf-8", "replace")
    if status in (-1, -2, -3):  # bad argument / unsupported / capacity
        raise ValueError(f"{what}: {msg} (status {status})")
    if status == IRIS_E_EPILOGUE_TIMEOUT:
        raise EpilogueTimeout(f"{what}: {msg} (status {status})")
    raise IrisError(f"{what}: {msg} (status {status})")
