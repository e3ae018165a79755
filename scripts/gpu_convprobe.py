"""Per-layer forward time of the CRNN's 14 convolutions (batch 64, c3 shape) in both memory formats, MIOpen NORMAL find:
which layers are slow, and would NCHW (Winograd / direct solvers) beat the NHWC implicit-GEMM kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
layers = [(1, 32, 64, 512), (32, 32, 64, 512), (32, 64, 32, 256), (64, 64, 32, 256), (64, 128, 16, 128), (128, 128, 16, 128),
          (128, 256, 8, 64), (256, 256, 8, 64), (256, 512, 4, 32), (512, 512, 4, 32)]
B = 64
for cin, cout, h, w in layers:
    res = []
    for fmt in (torch.channels_last, torch.contiguous_format):
        x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=fmt)
        wt = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=fmt)
        for _ in range(3):
            torch.nn.functional.conv2d(x, wt, None, padding=1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            y = torch.nn.functional.conv2d(x, wt, None, padding=1)
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 10 * 1e3)
    gf = 2 * 9 * cin * cout * h * w * B / 1e9
    print(f"conv {cin:3d}->{cout:3d} @{h}x{w}: NHWC {res[0]:7.1f} us ({gf / res[0] * 1e-3:6.1f} TFLOP/s) | NCHW {res[1]:7.1f} us ({gf / res[1] * 1e-3:6.1f} TFLOP/s)", flush=True)
