"""the generators' image heads (conv3x3 ngf -> 3, model.py:464-475): value against fp64 and times of forward / data gradient / weight gradient"""
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
for (B, C, H) in [(16, 48, 256), (16, 48, 128), (16, 48, 64), (3, 20, 40)]:
    x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(3, C, 3, 3, device="cuda") * 0.1
    y = ops.conv2d_forward(x, w, 1, 1, 1, 0); dy = torch.randn_like(y)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    e = float((y.double() - ref).norm() / ref.norm())
    g = torch.zeros_like(w)
    print("B%d %d->3 %dx%d: fwd rel-L2 %.2e | fwd %.1f us  dgrad %.1f us  wgrad %.1f us" % (B, C, H, H, e,
          t(lambda: ops.conv2d_forward(x, w, 1, 1, 1, 0)), t(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0)),
          t(lambda: ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 0, out=g, accumulate=True))), flush=True)
