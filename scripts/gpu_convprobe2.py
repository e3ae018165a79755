"""Block 1 of the CRNN (conv 1->32, conv 32->32 at 64x512, batch 64): forward and backward time per memory format."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
B = 64
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for cin, cout, h, w in [(1, 32, 64, 512), (32, 32, 64, 512), (32, 64, 32, 256), (64, 64, 32, 256)]:
    for name, fmt in (("NHWC", torch.channels_last), ("NCHW", torch.contiguous_format)):
        x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
        wt = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
        y = torch.nn.functional.conv2d(x, wt, None, padding=1)
        g = torch.randn_like(y)
        fwd = t(lambda: torch.nn.functional.conv2d(x, wt, None, padding=1))
        def bwd():
            yy = torch.nn.functional.conv2d(x, wt, None, padding=1)
            gi, gw = torch.autograd.grad(yy, (x, wt), g)
        both = t(bwd)
        print(f"conv {cin:3d}->{cout:3d} @{h}x{w} {name}: fwd {fwd:7.1f} us, fwd+bwd {both:7.1f} us, bwd {both - fwd:7.1f} us", flush=True)
