"""Round-6 microbenchmark: the Winograd convolution with its GEMMs on the BF16 matrix cores through a three-term
split of both operands (challenge_amd/csrc/k_conv_wino_b3.h) against the exact-fp32 Winograd kernel (k_conv_wino.h), both built alone
as scripts/microbench/libwino.so: error against fp64 side by side (the gate: <= 1.5x the fp32 kernel's, <= 1e-6 of the peak) and time
on the CRNN's twelve layers at batch 64.   usage: gpu_wino_b3_bench.py [check|time|all]"""
import ctypes as C
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gpu_wino_bench as W  # noqa: E402  (pack / wino / to_chunked / from_chunked / timeit of the fp32 kernel; same library)

lib, dev = W.lib, W.dev
lib.iris_wino_b3_packed_len.restype = C.c_size_t
lib.iris_wino_b3_packed_len.argtypes = [C.c_int, C.c_int]
lib.iris_wino_b3_pack_weights_device.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.iris_conv3x3_wino_b3.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]


def stream():
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def pack_b3(w, transposed=False):
    co, ci = w.shape[:2]
    cin, cout = (co, ci) if transposed else (ci, co)
    out = torch.empty(lib.iris_wino_b3_packed_len(cin, cout), dtype=torch.float32, device=dev)
    so, si, sh, sw = w.stride()
    rc = lib.iris_wino_b3_pack_weights_device(w.data_ptr(), so, si, sh, sw, cin, cout, 1 if transposed else 0, out.data_ptr(), stream())
    assert rc == 0, lib.wino_last_error()
    return out


def wino_b3(x, packed, bias, cout, pool, out_nhwc=False, in_nhwc=False, relu=True):
    if in_nhwc:
        b, h, w, cin = x.shape
    else:
        b, cbk, h, w, _ = x.shape
        cin = 8 * cbk
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout) if out_nhwc else (b, cout // 8, ho, wo, 8), device=dev)
    rc = lib.iris_conv3x3_wino_b3(x.data_ptr(), packed.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), b, h, w, cin, cout,
                                  (1 if pool else 0) | (2 if out_nhwc else 0) | (4 if in_nhwc else 0) | (8 if relu else 0), stream())
    assert rc == 0, lib.wino_last_error()
    return y


def check(b, h, w, cin, cout, pool, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(b, h, w, cin, generator=g, device=dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g, device=dev) * 0.1
    xc = W.to_chunked(x)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=1).relu()
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True)
    ref = ref.permute(0, 2, 3, 1)
    peak = float(ref.abs().max())
    y32 = W.from_chunked(W.wino(xc, W.pack(wt), bias, cout, pool), cout)
    yb3 = W.from_chunked(wino_b3(xc, pack_b3(wt), bias, cout, pool), cout)
    yb3n = wino_b3(xc, pack_b3(wt), bias, cout, pool, out_nhwc=True)
    yb3i = wino_b3(x.contiguous(), pack_b3(wt), bias, cout, pool, out_nhwc=True, in_nhwc=True)
    same = bool(torch.equal(yb3, yb3n) and torch.equal(yb3n, yb3i))
    e32, eb3 = float((y32.double() - ref).abs().max()) / peak, float((yb3.double() - ref).abs().max()) / peak
    # root-mean-square error too: a truncation-like bias would show there before it shows in the maximum
    r32, rb3 = float((y32.double() - ref).pow(2).mean().sqrt()) / peak, float((yb3.double() - ref).pow(2).mean().sqrt()) / peak
    return e32, eb3, r32, rb3, same


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    ok = True
    if which in ("check", "all"):
        print("error against fp64, max |y - ref| / max |ref| (rms in brackets): exact-fp32 Winograd kernel | bf16 x 3 kernel | ratio | layouts agree")
        for shp in [(2, 8, 12, 16, 64, False), (3, 7, 9, 16, 64, False), (3, 7, 9, 32, 128, True), (2, 16, 128, 128, 128, False),
                    (2, 16, 128, 128, 128, True), (2, 4, 32, 512, 512, True), (1, 5, 33, 48, 64, True), (2, 32, 256, 32, 64, False),
                    (3, 2, 3, 64, 64, True), (2, 8, 64, 256, 256, False)]:
            e32, eb3, r32, rb3, same = check(*shp)
            good = eb3 <= max(1.5 * e32, 3e-7) and eb3 <= 1e-6 and same
            ok = ok and good
            print(f"  B {shp[0]} {shp[1]}x{shp[2]} {shp[3]}->{shp[4]} pool {int(shp[5])}: {e32:.2e} ({r32:.1e}) | {eb3:.2e} ({rb3:.1e}) | x{eb3 / e32:.2f} | {same}"
                  + ("" if good else "   <-- GATE"), flush=True)
        print("gate (<= 1.5x the exact-fp32 kernel's error [or 3e-7], <= 1e-6 of the peak, layouts bit-identical):", "PASS" if ok else "FAIL")
    if which in ("time", "all"):
        print("timing, batch 64, chunked in / out (us):  exact-fp32 Winograd | bf16 x 3 | speed-up | bf16x3 TF of Winograd multiplies")
        rows = [(32, 256, 32, 64, False), (32, 256, 64, 64, False), (32, 256, 64, 64, True),
                (16, 128, 64, 128, False), (16, 128, 128, 128, False), (16, 128, 128, 128, True),
                (8, 64, 128, 256, False), (8, 64, 256, 256, False), (8, 64, 256, 256, True),
                (4, 32, 256, 512, False), (4, 32, 512, 512, False), (4, 32, 512, 512, True)]
        if os.environ.get("WINO_ROWS") == "short":
            rows = [(16, 128, 128, 128, False), (4, 32, 512, 512, False)]
        t32 = tb3 = 0.0
        for h, w, cin, cout, pool in rows:
            g = torch.Generator(device=dev).manual_seed(1)
            x = torch.randn(64, h, w, cin, generator=g, device=dev)
            wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
            bias = torch.zeros(cout, device=dev)
            xch = W.to_chunked(x)
            pk, pk3 = W.pack(wt), pack_b3(wt)
            a = W.timeit(lambda: W.wino(xch, pk, bias, cout, pool))
            b_ = W.timeit(lambda: wino_b3(xch, pk3, bias, cout, pool))
            n = W.timeit(lambda: wino_b3(x, pk3, bias, cout, pool, out_nhwc=True, in_nhwc=True, relu=False)) if not pool else float("nan")
            gf = 2.0 * 64 * h * w * cin * cout * 9 / 1e9
            t32 += a
            tb3 += b_
            print(f"  {h}x{w} {cin}->{cout} pool {int(pool)}: {a:7.1f} | {b_:7.1f} | x{a / b_:.2f} | {gf / b_ * 1e3 / 2.25:6.1f} TF (fp32-equivalent)"
                  f" | channels-last in / out, bare: {n:7.1f}", flush=True)
        print(f"  sum of the 12 layers: exact-fp32 {t32:.0f} us, bf16 x 3 {tb3:.0f} us (x{t32 / tb3:.2f})")
    sys.exit(0 if ok else 1)
