"""mogan_dconv2.hip against fp64 torch: forward / data gradient of 3x3 s1 p1 and the data gradient of 4x4 s2 p1 (2x2 parity
sub-convolutions) on 8 x 32 tile grids, plus the weight gradients of the same shapes; run with MOGAN_WINO=0 so that the 3x3 shapes
reach the direct kernels."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
torch.manual_seed(0)
CASES = [  # B, Cin, H, W, Cout, k, s
    (2, 96, 64, 64, 192, 3, 1), (2, 96, 32, 64, 96, 3, 1), (3, 32, 8, 32, 64, 3, 1), (2, 64, 12, 32, 160, 3, 1), (1, 32, 4, 32, 128, 3, 1), (2, 48, 16, 32, 80, 3, 1), (1, 384, 32, 32, 384, 3, 1),
    (2, 96, 128, 128, 192, 4, 2), (2, 192, 64, 64, 384, 4, 2), (2, 64, 16, 64, 96, 4, 2), (2, 80, 16, 64, 128, 4, 2), (1, 16, 16, 64, 768, 4, 2),
    # 16 x 16 tiles (maps with 16-pixel rows) and the space-to-depth forward of 4x4 s2
    (2, 384, 32, 32, 768, 4, 2), (3, 64, 32, 32, 96, 4, 2), (2, 24, 32, 64, 100, 4, 2), (2, 64, 16, 16, 128, 3, 1), (2, 32, 32, 16, 64, 3, 1), (1, 8, 64, 32, 64, 4, 2), (16, 96, 128, 128, 192, 4, 2)]
worst = 0.0
for (B, Cin, H, W, Cout, k, s) in CASES:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    xd, wd = x.double().requires_grad_(True), w.double()
    yd = F.conv2d(xd, wd, None, s, 1)
    gy = torch.randn_like(yd)
    wdd = wd.clone().requires_grad_(True)
    F.conv2d(xd.detach(), wdd, None, s, 1).backward(gy)
    yd.backward(gy)
    y = ops.conv2d_forward(x, w, s, 1, 1, 0)
    dx = ops.conv2d_dgrad(gy.float(), w, x.shape, s, 1, 1, 0)
    dw = ops.conv2d_wgrad(gy.float(), x, w.shape, s, 1, 1, 0)
    g2 = torch.full_like(w, 0.5); ops.conv2d_wgrad(gy.float(), x, w.shape, s, 1, 1, 0, out=g2, accumulate=True)
    torch.cuda.synchronize()
    ew = float((dw.double() - wdd.grad).norm() / wdd.grad.norm()); ew2 = float((g2.double() - 0.5 - wdd.grad).norm() / wdd.grad.norm())
    worst = max(worst, ew, ew2)
    ef = float((y.double() - yd).norm() / yd.norm()); eb = float((dx.double() - xd.grad).norm() / xd.grad.norm())
    worst = max(worst, ef, eb)
    print("B%d %3d->%3d %3dx%-3d k%d s%d  fwd rel-L2 %.2e  dgrad rel-L2 %.2e  wgrad %.2e (accumulating %.2e)" % (B, Cin, Cout, H, W, k, s, ef, eb, ew, ew2), flush=True)
print("worst %.2e %s" % (worst, "OK" if worst < 5e-6 else "FAIL"))
