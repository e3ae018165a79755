"""Round-5 microbenchmark: the weight gradient of the CRNN's 3x3 convolutions (blocks 2-5) as Winograd F(2x2, 3x3) on the fp32
matrix cores (challenge_amd/csrc/k_conv_wino_wrw.h, built alone into scripts/microbench/libwino.so) against MIOpen's
weight-gradient kernels on the same shapes (channels-last fp32, batch 64 at the training geometry)."""
import ctypes as C
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from challenge_amd import sj_train as S  # noqa: E402
S.configure_miopen()
lib = C.CDLL(os.environ.get("WINO_LIB", os.path.join(ROOT, "scripts", "microbench", "libwino.so")))
lib.iris_wino_wrw_workspace_len.restype = C.c_size_t
lib.iris_wino_wrw_workspace_len.argtypes = [C.c_int] * 5
lib.iris_conv3x3_wino_wrw.argtypes = [C.c_void_p] * 3 + [C.c_long] * 4 + [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p]
lib.wino_last_error.restype = C.c_char_p
dev = torch.device("cuda", 0)


def wrw(x, dy, dw=None, accumulate=False, ws=None):
    """x: [B, H, W, Cin], dy: [B, H, W, Cout] (contiguous) -> dW [Cout, Cin, 3, 3] (channels_last strides)."""
    b, h, w, cin = x.shape
    cout = dy.shape[3]
    if dw is None:
        dw = torch.empty(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    n = lib.iris_wino_wrw_workspace_len(b, h, w, cin, cout)
    if ws is None:
        ws = torch.empty(n, device=dev)
    so, si, sh, sw = dw.stride()
    rc = lib.iris_conv3x3_wino_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), so, si, sh, sw, b, h, w, cin, cout, int(accumulate),
                                   ws.data_ptr(), ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0, lib.wino_last_error()
    return dw


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ref_dw(x, dy, dtype):
    xn, dn = x.permute(0, 3, 1, 2).to(dtype), dy.permute(0, 3, 1, 2).to(dtype)
    w = torch.zeros(dy.shape[3], x.shape[3], 3, 3, device=dev, dtype=dtype)
    return torch.ops.aten.convolution_backward(dn, xn, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


def check(b, h, w, cin, cout, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(b, h, w, cin, generator=g, device=dev)
    dy = torch.randn(b, h, w, cout, generator=g, device=dev)
    got = wrw(x, dy)
    ref = ref_dw(x, dy, torch.float64)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    err_mi = float((ref_dw(x, dy, torch.float32).double() - ref).abs().max() / ref.abs().max())
    again = wrw(x, dy)
    assert torch.equal(got, again), "not deterministic"
    acc = wrw(x, dy, dw=got.clone(), accumulate=True)
    assert float((acc - 2 * got).abs().max()) <= 1e-6 * float(got.abs().max()), "accumulate"
    return err, err_mi


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("correctness (max |dW - fp64| / max |fp64|; MIOpen fp32 beside it):")
    for shp in [] if which == "time" else [(2, 8, 12, 64, 64), (3, 7, 9, 64, 128), (1, 5, 33, 128, 64), (2, 16, 128, 128, 128),
                                           (2, 4, 32, 512, 512), (64, 4, 32, 256, 512), (5, 1, 1, 64, 64), (1, 2, 3, 64, 64), (2, 9, 14, 32, 64), (3, 6, 8, 96, 128), (2, 9, 14, 32, 32), (3, 6, 8, 64, 96)]:
        err, err_mi = check(*shp)
        print(f"  B {shp[0]} {shp[1]}x{shp[2]} {shp[3]}->{shp[4]}: wino {err:.2e}  miopen {err_mi:.2e}", flush=True)
        assert err < 2e-5, err
    if which == "check":
        sys.exit(0)
    print("timing, batch 64 (us; GFLOP of the direct weight gradient):")
    rows = [(64, 512, 32, 32), (32, 256, 32, 64), (32, 256, 64, 64), (16, 128, 64, 128), (16, 128, 128, 128), (8, 64, 128, 256), (8, 64, 256, 256),
            (4, 32, 256, 512), (4, 32, 512, 512)]
    if os.environ.get("WINO_ROWS") == "short":
        rows = [(16, 128, 128, 128), (4, 32, 512, 512)]
    tot_w = tot_m = 0.0
    for h, w, cin, cout in rows:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(64, h, w, cin, generator=g, device=dev)
        dy = torch.randn(64, h, w, cout, generator=g, device=dev)
        dw = torch.empty(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        ws = torch.empty(lib.iris_wino_wrw_workspace_len(64, h, w, cin, cout), device=dev)
        xn, dn = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
        wz = torch.zeros(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        t_w = timeit(lambda: wrw(x, dy, dw=dw, ws=ws))
        t_m = timeit(lambda: torch.ops.aten.convolution_backward(dn, xn, wz, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                 [False, True, False]))
        gf = 2.0 * 64 * h * w * cin * cout * 9 / 1e9
        tot_w += t_w
        tot_m += t_m
        print(f"  {h}x{w} {cin}->{cout}: wino {t_w:7.1f} ({gf / t_w * 1e3 / 2.25:6.1f} TF on the MFMA; {ws.numel() * 4 / 2**20:5.0f} MiB of partials) | "
              f"miopen {t_m:7.1f} ({gf / t_m * 1e3:6.1f} TF) | x{t_m / t_w:.2f}", flush=True)
    print(f"  sum: wino {tot_w:.0f} us, miopen {tot_m:.0f} us")
