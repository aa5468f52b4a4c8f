"""Diagnostic: fused BN+activation forward/backward at full-size shapes against fp64 torch-CPU autograd."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, rel_l2
load_pkg()
from mogan_amd.hip import ops
CASES = [(4, 192, 64, ops.ACT_LRELU), (4, 384, 32, ops.ACT_LRELU), (16, 192, 64, ops.ACT_LRELU), (4, 192, 128, ops.ACT_GLU),
         (16, 96, 256, ops.ACT_GLU), (4, 96, 64, ops.ACT_NONE), (4, 768, 16, ops.ACT_LRELU)]
for B, C, H, act in CASES:
    for mean_shift in (0.0, 3.0):
        g = torch.Generator().manual_seed(B * C + H)
        x = torch.randn(B, C, H, H, generator=g) * 1.7 + mean_shift
        gam = torch.randn(C, generator=g) * 0.1 + 1.0; bet = torch.randn(C, generator=g) * 0.1
        Cy = C // 2 if act == ops.ACT_GLU else C
        dy = torch.randn(B, Cy, H, H, generator=g)
        xd, gd, bd = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
        t = F.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5)
        if act == ops.ACT_LRELU: yd = F.leaky_relu(t, 0.2)
        elif act == ops.ACT_GLU: yd = t[:, :Cy] * torch.sigmoid(t[:, Cy:])
        else: yd = t
        yd.backward(dy.double())
        xg = x.cuda().requires_grad_(True); gg = gam.cuda().requires_grad_(True); bg = bet.cuda().requires_grad_(True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y = ops.bn_act(xg, gg, bg, rm, rv, act, 0.2)
        y.backward(dy.cuda()); torch.cuda.synchronize()
        print("B=%2d C=%3d %3dx%-3d act %d shift %.0f: y %.2e  dx %.2e  dgamma %.2e  dbeta %.2e"
              % (B, C, H, H, act, mean_shift, rel_l2(y, yd), rel_l2(xg.grad, xd.grad), rel_l2(gg.grad, gd.grad), rel_l2(bg.grad, bd.grad)), flush=True)
