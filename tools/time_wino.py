"""3x3 s1 p1 forward / data gradient through the C ABI on the ResBlock shapes: run once with MOGAN_WINO=1 (fused Winograd
F(2x2,3x3), csrc/mogan_wino.hip) and once with MOGAN_WINO=0 (direct halo-tile kernel).  TF = direct-convolution flops / time."""
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
out = []
for (B, Cin, H, Cout) in [(16, 16, 128, 96), (16, 32, 128, 96), (16, 48, 128, 96), (16, 96, 128, 192), (16, 96, 128, 96),
                          (16, 96, 64, 192), (16, 96, 64, 96), (16, 192, 64, 96), (24, 768, 64, 768)]:
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    y = ops.conv2d_forward(x, w, 1, 1, 1, 0); dy = torch.randn_like(y)
    ref = torch.nn.functional.conv2d(x[:2].double(), w.double(), None, 1, 1)
    err = (y[:2].double() - ref).abs().max().item() / ref.abs().max().item()
    tf, td = t(lambda: ops.conv2d_forward(x, w, 1, 1, 1, 0)), t(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0))
    gf = 2.0 * B * H * H * Cout * Cin * 9 / 1e9
    print("B%d %3d->%3d %dx%d: fwd %.3f ms %5.0f TF  dgrad %.3f ms %5.0f TF  (fwd err %.1e)" % (B, Cin, Cout, H, H, tf, gf / tf, td, gf / td, err), flush=True)
