"""Diagnostic: fwd / dgrad / wgrad of the D down-convolutions at full size against fp64 torch-CPU convolutions.
Usage: python tools/diag_conv_precision.py [B ...]"""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, rel_l2
load_pkg()
from mogan_amd.hip import ops
LAYERS = [(3, 256, 96, 4, 2, 1), (96, 128, 192, 4, 2, 1), (192, 64, 384, 4, 2, 1), (384, 32, 768, 4, 2, 1),
          (96, 64, 192, 3, 1, 1), (96, 128, 96, 3, 1, 1)]
for B in [int(a) for a in sys.argv[1:]] or [4, 16]:
    for Cin, H, Cout, k, s, p in LAYERS:
        g = torch.Generator().manual_seed(Cin * 7 + H)
        x = torch.randn(B, Cin, H, H, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
        xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
        yd = F.conv2d(xd, wd, None, s, p)
        dy = torch.randn(yd.shape, generator=g)
        yd.backward(dy.double())
        xg, wg, dyg = x.cuda(), w.cuda(), dy.cuda()
        y = ops.conv2d_forward(xg, wg, s, p, p, 0)
        dx = ops.conv2d_dgrad(dyg, wg, xg.shape, s, p, p, 0)
        dw = ops.conv2d_wgrad(dyg, xg, wg.shape, s, p, p, 0)
        torch.cuda.synchronize()
        # where is the dgrad error? interior vs border
        e = (dx.cpu().double() - xd.grad).abs()
        border = torch.ones_like(e, dtype=torch.bool); border[:, :, 2:-2, 2:-2] = False
        print("B=%2d %4d->%4d %3dx%-3d k%d s%d: fwd %.2e  dgrad %.2e (max err interior %.2e border %.2e)  wgrad %.2e"
              % (B, Cin, Cout, H, H, k, s, rel_l2(y, yd), rel_l2(dx, xd.grad), float(e[~border].max()), float(e[border].max()),
                 rel_l2(dw, wd.grad)), flush=True)
