"""Timing of the <=4-channel convolutions (image heads, first D convolution) through the C ABI."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, Cin, H, Cout, k, s) in [(16, 48, 256, 3, 3, 1), (16, 48, 128, 3, 3, 1), (16, 3, 256, 96, 4, 2), (16, 3, 128, 96, 4, 2)]:
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    y = ops.conv2d_forward(x, w, s, 1, 1, 0); dy = torch.randn_like(y)
    tf = t(lambda: ops.conv2d_forward(x, w, s, 1, 1, 0)); td = t(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, 0))
    g = torch.zeros_like(w)
    tw = t(lambda: ops.conv2d_wgrad(dy, x, w.shape, s, 1, 1, 0, out=g, accumulate=True))
    mb = (x.numel() + y.numel()) * 4 / 1e6
    print("B%d %d->%d %dx%d k%d s%d: fwd %.3f ms (%.0f GB/s)  dgrad %.3f ms (%.0f GB/s)  wgrad %.3f ms (%.0f GB/s)"
          % (B, Cin, Cout, H, H, k, s, tf, mb / tf, td, mb / td, tw, mb / tw))
