"""Wall-time microbench of the big conv layers (fwd / dgrad / wgrad) through the C ABI."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
CASES = [  # B, Cin, Hs, Ws, Cout, k, stride, up
    (16, 96, 128, 128, 96, 3, 1, 1), (16, 96, 128, 128, 192, 3, 1, 0), (16, 96, 128, 128, 96, 3, 1, 0),
    (16, 96, 64, 64, 192, 3, 1, 0), (16, 384, 16, 16, 384, 3, 1, 1),
    (16, 96, 128, 128, 192, 4, 2, 0), (16, 192, 64, 64, 384, 4, 2, 0), (16, 384, 32, 32, 768, 4, 2, 0),
    (16, 768, 16, 16, 1536, 4, 2, 0), (16, 1536, 8, 8, 3072, 4, 2, 0), (16, 3072, 4, 4, 1536, 3, 1, 0)]
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, Cin, Hs, Ws, Cout, k, s, up) in CASES:
    x = torch.randn(B, Cin, Hs, Ws, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.02
    y = ops.conv2d_forward(x, w, s, 1, 1, up); dy = torch.randn_like(y)
    gf = 2.0 * y.numel() * Cin * k * k / 1e9
    g = torch.zeros_like(w)
    tf = t(lambda: ops.conv2d_forward(x, w, s, 1, 1, up))
    td = t(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, up))
    tw = t(lambda: ops.conv2d_wgrad(dy, x, w.shape, s, 1, 1, up, out=g, accumulate=True))
    print("B%d %4d->%4d %3dx%-3d k%d s%d up%d  %6.1f GF | fwd %6.3f ms %6.1f TF | dgrad %6.3f ms %6.1f TF | wgrad %6.3f ms %6.1f TF"
          % (B, Cin, Cout, Hs, Ws, k, s, up, gf, tf, gf / tf, td, gf / td, tw, gf / tw), flush=True)
