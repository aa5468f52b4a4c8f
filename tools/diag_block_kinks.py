"""Diagnostic: downBlock(384,768) at (16,384,32,32) -- the loaded library's dx / dW against fp64, plain and with the LeakyReLU
decisions of the run under test imposed on the fp64 evaluation (mask-matched).  If the plain error is ~1e-4 and the
mask-matched one ~1e-6, the difference to the fixture is sign decisions of pre-activations within rounding of zero."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import det_array, det_fill_state, load_pkg, rel_l2
load_pkg()
from mogan_amd.attngan import model

torch.manual_seed(0)
mod = model.downBlock(384, 768)
sd = det_fill_state(mod, "fw.down.")
x = torch.from_numpy(det_array("fw.down.x", (16, 384, 32, 32)))
g = torch.from_numpy(det_array("fw.down.g", (16, 768, 16, 16)))
modg = mod.cuda().train()
xg = x.cuda().requires_grad_(True)
y = modg(xg)
y.backward(g.cuda())
torch.cuda.synchronize()
w, gam, bet = (sd[k].double() for k in ("0.weight", "1.weight", "1.bias"))


def f64(mask=None):
    xd = x.double().requires_grad_(True)
    wd = w.clone().requires_grad_(True)
    t = F.batch_norm(F.conv2d(xd, wd, None, 2, 1), None, None, gam, bet, True, 0.1, 1e-5)
    pos = (t > 0) if mask is None else mask
    out = torch.where(pos, t, 0.2 * t)
    out.backward(g.double())
    return t.detach(), out.detach(), xd.grad, wd.grad


t64, y64, dx64, dw64 = f64()
mask = (y.detach().cpu() > 0)
flips = int((mask != (t64 > 0)).sum())
print("library:", os.environ.get("MOGAN_LIB", "default"))
print("sign decisions that differ from fp64: %d of %d; |t64| at those: max %.2e" % (
    flips, mask.numel(), float(t64.abs()[mask != (t64 > 0)].max()) if flips else 0.0))
print("plain        : y %.2e  dx %.2e  dW %.2e" % (rel_l2(y.detach(), y64), rel_l2(xg.grad, dx64), rel_l2(modg[0].weight.grad, dw64)))
_, ym, dxm, dwm = f64(mask)
print("mask-matched : y %.2e  dx %.2e  dW %.2e" % (rel_l2(y.detach(), ym), rel_l2(xg.grad, dxm), rel_l2(modg[0].weight.grad, dwm)))
