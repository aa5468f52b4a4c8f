"""4x4 s2 p1 convolutions of the discriminators through the C ABI: forward / data gradient / weight gradient.
Run with MOGAN_WINO22=1 (fused Winograd F(2x2,2x2), csrc/mogan_wino22.hip) and =0.  TF = direct-convolution flops / time."""
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
for (B, Cin, H, Cout) in [(16, 96, 128, 192), (16, 192, 64, 384), (16, 384, 32, 768), (16, 768, 16, 1536), (16, 1536, 8, 3072),
                          (16, 192, 32, 384), (16, 384, 16, 768), (16, 768, 8, 1536), (16, 384, 16, 384), (16, 384, 8, 768)]:
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.02
    y = ops.conv2d_forward(x, w, 2, 1, 1, 0); dy = torch.randn_like(y); g = torch.zeros_like(w)
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), None, 2, 1)
    err = (y[:1].double() - ref).abs().max().item() / ref.abs().max().item()
    tf = t(lambda: ops.conv2d_forward(x, w, 2, 1, 1, 0)); td = t(lambda: ops.conv2d_dgrad(dy, w, x.shape, 2, 1, 1, 0))
    tw = t(lambda: ops.conv2d_wgrad(dy, x, w.shape, 2, 1, 1, 0, out=g, accumulate=True))
    gf = 2.0 * y.numel() * Cin * 16 / 1e9
    print("B%d %4d->%4d %3dx%-3d: fwd %.3f ms %5.0f TF | dgrad %.3f ms %5.0f TF | wgrad %.3f ms %5.0f TF  (fwd err %.1e)"
          % (B, Cin, Cout, H, H, tf, gf / tf, td, gf / td, tw, gf / tw, err), flush=True)
