"""Winograd F(2x2,3x3) forward / data gradient / weight gradient through the C ABI: values against fp64 torch on the shapes of the
generator's ResBlocks (+ ragged / small ones), and wall time per layer.  MOGAN_LIB selects a lab build."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
import torch.nn.functional as F
dev = "cuda"
CASES = [(16, 96, 128, 128, 192), (16, 96, 128, 128, 96), (16, 192, 128, 128, 96), (16, 96, 64, 64, 192), (16, 96, 64, 64, 96),
         (16, 192, 64, 64, 96), (3, 64, 36, 64, 96), (2, 32, 20, 96, 160), (5, 48, 64, 32, 64)]
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def rel(a, b):
    return float((a.double() - b).norm() / b.norm())
worst = 0.0
for (B, Cin, H, W, Cout) in CASES:
    g = torch.Generator(device=dev).manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, device=dev, generator=g); w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) * 0.05
    dy = torch.randn(B, Cout, H, W, device=dev, generator=g)
    y = ops.conv2d_forward(x, w, 1, 1, 1, 0)
    dx = ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0)
    dw = torch.zeros_like(w); ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 0, out=dw, accumulate=True)
    xd, wd, dyd = x.double().requires_grad_(True), w.double().requires_grad_(True), dy.double()
    yr = F.conv2d(xd, wd, padding=1); yr.backward(dyd)
    e = (rel(y, yr.detach()), rel(dx, xd.grad), rel(dw, wd.grad))
    worst = max(worst, *e)
    gf = 2.0 * y.numel() * Cin * 9 / 1e9
    tf = t(lambda: ops.conv2d_forward(x, w, 1, 1, 1, 0)); td = t(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0))
    tw = t(lambda: ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 0, out=dw, accumulate=True))
    print("B%-2d %4d->%4d %3dx%-3d %6.1f GF | fwd %6.1f us %6.1f TF %.1e | dgrad %6.1f us %6.1f TF %.1e | wgrad %6.1f us %6.1f TF %.1e"
          % (B, Cin, Cout, H, W, gf, tf * 1e3, gf / tf, e[0], td * 1e3, gf / td, e[1], tw * 1e3, gf / tw, e[2]), flush=True)
print("worst rel-L2 %.2e %s" % (worst, "OK" if worst < 5e-6 else "FAIL"))
