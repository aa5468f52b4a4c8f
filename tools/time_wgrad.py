"""wgrad-only microbench (dconv_wgrad_kernel) on the big 3x3 layers; MOGAN_LIB selects a lab variant."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
CASES = [(16, 96, 128, 128, 96, 3, 1, 1), (16, 96, 128, 128, 192, 3, 1, 0), (16, 96, 64, 64, 192, 3, 1, 0),
         (16, 192, 64, 64, 384, 4, 2, 0)]
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for (B, Cin, Hs, Ws, Cout, k, s, up) in CASES:
    x = torch.randn(B, Cin, Hs, Ws, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.02
    y = ops.conv2d_forward(x, w, s, 1, 1, up); dy = torch.randn_like(y)
    gf = 2.0 * y.numel() * Cin * k * k / 1e9
    g = torch.zeros_like(w)
    tw = t(lambda: ops.conv2d_wgrad(dy, x, w.shape, s, 1, 1, up, out=g, accumulate=True))
    out.append("%.1f" % (gf / tw))
print(os.environ.get("MOGAN_LIB", "product").split("/")[-1], "wgrad TF:", " ".join(out))
