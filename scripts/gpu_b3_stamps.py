"""Where a work item of the bf16x3 Winograd kernel spends its time: s_memrealtime / s_memtime stamps of wave 0 at six points
(start, prologue done, K loop done, exchange written, exchange barrier passed, stores issued) - libwino_stamps.so
(hipcc ... -DIRIS_B3_STAMPS=1 -o scripts/microbench/libwino_stamps.so scripts/microbench/wino_conv.hip)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
os.environ["WINO_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "microbench", "libwino_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_wino_b3_bench as B  # noqa: E402
W, lib, dev = B.W, B.lib, B.dev
lib.iris_b3_read_stamps.argtypes = [C.c_void_p]
for h, w, cin, cout, pool in [(16, 128, 128, 128, False), (4, 32, 512, 512, False), (32, 256, 32, 64, False), (8, 64, 256, 256, True)]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(64, h, w, cin, generator=g, device=dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.zeros(cout, device=dev)
    xch, pk3 = W.to_chunked(x), B.pack_b3(wt)
    for _ in range(5):
        B.wino_b3(xch, pk3, bias, cout, pool)
    torch.cuda.synchronize()
    st = np.zeros(256 * 2 * 10 * 2, np.uint64)
    assert lib.iris_b3_read_stamps(st.ctypes.data) == 0
    st = st.reshape(256, 2, 10, 2).astype(np.float64)
    rt, cy = st[..., 0], st[..., 1]           # 100 MHz ticks, shader cycles
    names = ["prologue", "K loop", "drain + column half + exchange writes", "exchange barrier", "row half + stores"]
    print(f"{h}x{w} {cin}->{cout} pool {int(pool)}: {cin // 16} chunks per work item")
    for item in range(2):
        ok = rt[:, item, 5] > 0
        if not ok.any():
            continue
        d_us = np.diff(rt[ok, item][:, :6], axis=1) / 100.0
        d_cy = np.diff(cy[ok, item][:, :6], axis=1)
        pro = rt[ok, item]
        print(f"    prologue: decode + zero + barrier {np.median(pro[:, 6] - pro[:, 0]) / 100:.2f} us; requests {np.median(pro[:, 7] - pro[:, 6]) / 100:.2f}; "
              f"wait + barrier {np.median(pro[:, 8] - pro[:, 7]) / 100:.2f}; row stage + first position {np.median(pro[:, 1] - pro[:, 8]) / 100:.2f}")
        line = "; ".join(f"{n} {np.median(d_us[:, k]):.2f} us ({np.median(d_cy[:, k]) / max(np.median(d_us[:, k]), 1e-9) / 1e3:.2f} GHz)" for k, n in enumerate(names))
        print(f"  item {item}: {line}; whole item {np.median(rt[ok, item, 5] - rt[ok, item, 0]) / 100.0:.2f} us")
    if (rt[:, 1, 0] > 0).any():
        ok = rt[:, 1, 0] > 0
        print(f"  gap item 0 end -> item 1 start: {np.median(rt[ok, 1, 0] - rt[ok, 0, 5]) / 100.0:.2f} us")
