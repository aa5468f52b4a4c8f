"""lab: where wave 0 of wino3_fwd_kernel spends its cycles (a build of mogan_wino.hip with shader-clock stamps at the segment
boundaries, MOGAN_LIB=build/lab_wino_seg.so; see DESIGN_LOG.md E)"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
L = lib.load()
fn = ctypes.CDLL(os.environ["MOGAN_LIB"]).mogan_lab_w3_segments
buf = (ctypes.c_ulonglong * 8)()
names = ["halo -> LDS + barrier", "first transform + barrier", "K loop", "output transform (LDS passes)", "stores", "tiles", "of the K loop: barrier wait"]
for (B, Cin, H, W, Cout, dg) in [(16, 96, 128, 128, 192, 0), (16, 96, 128, 128, 192, 1), (16, 96, 128, 128, 96, 0), (16, 96, 64, 64, 192, 0)]:
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    y = ops.conv2d_forward(x, w, 1, 1, 1, 0); dy = torch.randn_like(y)
    f = (lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0)) if dg else (lambda: ops.conv2d_forward(x, w, 1, 1, 1, 0))
    f(); torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    fn(buf, 1)
    tot = sum(buf[i] for i in range(5))
    print("B%d %d->%d %dx%d %s: %.3f ms; wave-0 cycles: %s | tiles %d, cycles/tile %.0f, barrier wait in the K loop %.1f%% of it" % (
        B, Cin, Cout, H, W, "dgrad" if dg else "fwd", e0.elapsed_time(e1),
        ", ".join("%s %.1f%%" % (names[i], 100.0 * buf[i] / tot) for i in range(5)), buf[5], tot / max(buf[5], 1), 100.0 * buf[6] / max(buf[2], 1)))
