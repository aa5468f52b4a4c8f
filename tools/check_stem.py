"""mogan_stem.hip (conv4x4 s2 p1 from 3 channels + LeakyReLU) against fp64 torch, and its time beside the implicit-GEMM kernel
(round 5: MOGAN_STEM=0; the switch is gone since round 6, the comparison stays on record in profiles/r05_ab.txt)."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
torch.manual_seed(0)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
worst = 0.0
for (B, H, W, Cout) in [(16, 256, 256, 96), (16, 128, 128, 96), (16, 64, 64, 96), (3, 64, 128, 48), (2, 64, 64, 128), (1, 2, 64, 32), (5, 66, 64, 100), (24, 256, 256, 96)]:
    x = torch.randn(B, 3, H, W, device="cuda"); w = torch.randn(Cout, 3, 4, 4, device="cuda") * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None, 2, 1), 0.2)
    y = ops.conv2d_lrelu(x, w, 2, 1, 0.2) if hasattr(ops, "conv2d_lrelu") else None
    if y is None:
        raise SystemExit("ops.conv2d_lrelu missing")
    y0 = ops.conv2d_forward(x, w, 2, 1, 1, 0)
    ref0 = F.conv2d(x.double(), w.double(), None, 2, 1)
    e = float((y.double() - ref).norm() / ref.norm()); e0 = float((y0.double() - ref0).norm() / ref0.norm())
    em = float((y.double() - ref).abs().max())
    worst = max(worst, e, e0)
    us = t(lambda: ops.conv2d_lrelu(x, w, 2, 1, 0.2))
    mb = (y.numel() + x.numel()) * 4 / 1e6
    print("B%d 3->%d %dx%d: rel-L2 %.2e (plain conv %.2e) max-abs %.2e | %.1f us, %.2f TB/s" % (B, Cout, H, W, e, e0, em, us, mb / us), flush=True)
print("worst %.2e %s" % (worst, "OK" if worst < 2e-6 else "FAIL"))
