"""The 32 -> 32 convolution of block 1 at c3 size (batch 64, 64 x 512): MIOpen (channels-last / contiguous) + the engine's
epilogue passes against iris_conv3x3_c32_bias_relu.  usage: python3 scripts/gpu_conv32.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import frontend as FE, sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
b, h, w = 64, 64, 512
x = torch.randn(b, 32, h, w, device=dev).contiguous(memory_format=torch.channels_last)
xc = x.contiguous()
wt = (torch.randn(32, 32, 3, 3, device=dev) * 0.1)
wcl = wt.contiguous(memory_format=torch.channels_last)
bias = torch.randn(32, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


flops = 2 * b * h * w * 32 * 32 * 9
for name, fn in [
    ("MIOpen channels-last conv only", lambda: torch.nn.functional.conv2d(x, wcl, None, padding=1)),
    ("MIOpen contiguous conv only", lambda: torch.nn.functional.conv2d(xc, wt, None, padding=1)),
    ("MIOpen contiguous conv + NCHW bias/ReLU/pool pass (the engine so far)",
     lambda: FE.bias_relu_maxpool_nchw(torch.nn.functional.conv2d(xc, wt, None, padding=1), bias)),
    ("MIOpen channels-last conv + bias/ReLU/pool pass", lambda: FE.bias_relu_maxpool(torch.nn.functional.conv2d(x, wcl, None, padding=1), bias)),
    ("HIP MFMA conv + bias + ReLU", lambda: FE.conv3x3_c32_bias_relu(x, wt, bias)),
    ("HIP MFMA conv + bias + ReLU + MaxPool", lambda: FE.conv3x3_c32_bias_relu(x, wt, bias, pool=True)),
]:
    us = timed(fn)
    print(f"{name:75s} {us:7.1f} us  ({flops / us / 1e6:6.1f} TFLOP/s)")
