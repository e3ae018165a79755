"""Round-5 microbenchmark: Conv2D 3x3 + bias + ReLU (+ MaxPool) of the CRNN's blocks 2-5 as Winograd
F(2x2, 3x3) on the fp32 matrix cores (challenge_amd/csrc/k_conv_wino.h, built alone as scripts/microbench/libwino.so) against
MIOpen's convolution on the same shapes (channels-last fp32, the engine's own call), batch 64 at the c3 geometry."""
import ctypes as C
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from challenge_amd import sj_train as S  # noqa: E402
S.configure_miopen()
lib = C.CDLL(os.environ.get("WINO_LIB", os.path.join(ROOT, "scripts", "microbench", "libwino.so")))
lib.iris_wino_packed_len.restype = C.c_size_t
lib.iris_wino_packed_len.argtypes = [C.c_int, C.c_int]
lib.iris_wino_pack_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.iris_conv3x3_wino.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
lib.wino_last_error.restype = C.c_char_p
dev = torch.device("cuda", 0)


def pack(w):
    cout, cin = w.shape[:2]
    host = np.ascontiguousarray(w.detach().cpu().numpy(), np.float32)
    out = np.empty(lib.iris_wino_packed_len(cin, cout), np.float32)
    rc = lib.iris_wino_pack_weights(host.ctypes.data, cin, cout, out.ctypes.data)
    assert rc == 0, lib.wino_last_error()
    return torch.from_numpy(out).to(dev)


ZEROS = torch.zeros(64, device=dev)


def to_chunked(x_nhwc):
    """[B, H, W, C] -> [B, C / 8, H, W, 8] (the layout the Winograd layers hand to each other)."""
    b, h, w, c = x_nhwc.shape
    return x_nhwc.view(b, h, w, c // 8, 8).permute(0, 3, 1, 2, 4).contiguous()


def from_chunked(y, cout):
    b, cb, h, w, _ = y.shape
    return y.permute(0, 2, 3, 1, 4).reshape(b, h, w, cout)


def wino(x_chunked, packed, bias, cout, pool, out_nhwc=False, in_nhwc=False):
    """x_chunked: [B, C / 8, H, W, 8], or with in_nhwc a contiguous [B, H, W, C] tensor (the training step's layout)."""
    if in_nhwc:
        b, h, w, cin = x_chunked.shape
    else:
        b, cbk, h, w, _ = x_chunked.shape
        cin = 8 * cbk
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout) if out_nhwc else (b, cout // 8, ho, wo, 8), device=dev)
    rc = lib.iris_conv3x3_wino(x_chunked.data_ptr(), packed.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w,
                               cin, cout, (1 if pool else 0) | (2 if out_nhwc else 0) | (4 if in_nhwc else 0) | 8,
                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0, lib.wino_last_error()
    return y


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def check(b, h, w, cin, cout, pool, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(b, h, w, cin, generator=g, device=dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g, device=dev) * 0.1
    xc = to_chunked(x)
    y = from_chunked(wino(xc, pack(wt), bias, cout, pool), cout)
    y2 = wino(xc, pack(wt), bias, cout, pool, out_nhwc=True)
    assert torch.equal(y, y2), "chunked and channels-last outputs differ"
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=1).relu()
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True)
    ref = ref.permute(0, 2, 3, 1)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    mi = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), wt, bias, padding=1).relu()
    if pool:
        mi = torch.nn.functional.max_pool2d(mi, 2, 2, ceil_mode=True)
    err_mi = float((mi.permute(0, 2, 3, 1).double() - ref).abs().max() / ref.abs().max())
    return err, err_mi


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("correctness (max |y - fp64| / max |fp64|; MIOpen fp32 beside it):")
    for shp in [] if which == "time" else [(2, 8, 12, 8, 64, False), (3, 7, 9, 16, 64, False), (3, 7, 9, 16, 128, True), (2, 16, 128, 128, 128, False),
                (2, 16, 128, 128, 128, True), (2, 4, 32, 512, 512, True), (1, 5, 33, 24, 64, True)]:
        err, err_mi = check(*shp)
        print(f"  B {shp[0]} {shp[1]}x{shp[2]} {shp[3]}->{shp[4]} pool {shp[5]}: wino {err:.2e}  miopen {err_mi:.2e}", flush=True)
        assert err < 2e-5, err
    if which == "check":
        sys.exit(0)
    print("timing, batch 64 (us; GFLOP of the direct convolution):")
    rows = [(32, 256, 32, 64, False), (32, 256, 64, 64, False), (32, 256, 64, 64, True),
            (16, 128, 64, 128, False), (16, 128, 128, 128, False), (16, 128, 128, 128, True),
            (8, 64, 128, 256, False), (8, 64, 256, 256, False), (8, 64, 256, 256, True),
            (4, 32, 256, 512, False), (4, 32, 512, 512, False), (4, 32, 512, 512, True)]
    if os.environ.get("WINO_ROWS") == "short":
        rows = [(16, 128, 128, 128, True), (4, 32, 512, 512, True)]
    tot_w = tot_m = 0.0
    for h, w, cin, cout, pool in rows:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(64, h, w, cin, generator=g, device=dev)
        wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
        bias = torch.zeros(cout, device=dev)
        pk = pack(wt)
        xn = x.permute(0, 3, 1, 2)  # NCHW view with channels_last strides
        wcl = wt.contiguous(memory_format=torch.channels_last)
        xch = to_chunked(x)
        t_w = timeit(lambda: wino(xch, pk, bias, cout, pool))
        t_n = timeit(lambda: wino(x, pk, bias, cout, pool, out_nhwc=True, in_nhwc=True)) if not pool else float("nan")
        t_i = timeit(lambda: wino(x, pk, bias, cout, pool, in_nhwc=True)) if not pool else float("nan")
        t_o = timeit(lambda: wino(xch, pk, bias, cout, pool, out_nhwc=True)) if not pool else float("nan")
        t_m = timeit(lambda: torch.nn.functional.conv2d(xn, wcl, None, padding=1))
        gf = 2.0 * 64 * h * w * cin * cout * 9 / 1e9
        tot_w += t_w
        tot_m += t_m
        print(f"  {h}x{w} {cin}->{cout} pool {int(pool)}: wino {t_w:7.1f} ({gf / t_w * 1e3 / 2.25:6.1f} TF on the MFMA) | miopen conv alone {t_m:7.1f} "
              f"({gf / t_m * 1e3:6.1f} TF) | x{t_m / t_w:.2f} | channels-last in / out {t_n:7.1f} (+{100 * (t_n / t_w - 1):4.1f} %), in only +{100 * (t_i / t_w - 1):4.1f} %, out only +{100 * (t_o / t_w - 1):4.1f} %", flush=True)
    print(f"  sum of the 12 layers: wino {tot_w:.0f} us, miopen convolutions alone {tot_m:.0f} us (+ its epilogue passes in the engine)")
