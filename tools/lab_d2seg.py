"""lab: where wave 0 of dconv2_fwd_kernel spends its cycles (build: tools/lab/mklab.sh l9 mogan_dconv2 -DDCONV2_LAB=9)"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
L = lib.load()
buf = (ctypes.c_ulonglong * 8)()
names = ["prologue", "loads+mfma issue", "barrier 1", "stage store", "barrier 2", "epilogue", "stages", "-"]
for (B, Cin, H, W, Cout, k, s) in [(16, 96, 128, 128, 192, 3, 1), (16, 96, 64, 64, 192, 3, 1), (16, 192, 64, 64, 384, 4, 2)]:
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    if k == 3:
        f = lambda: ops.conv2d_forward(x, w, s, 1, 1, 0)
    else:
        y = ops.conv2d_forward(x, w, s, 1, 1, 0); dy = torch.randn_like(y)
        f = lambda: ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, 0)
    f(); torch.cuda.synchronize()
    L.mogan_lab_d2_segments(buf, 1)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    L.mogan_lab_d2_segments(buf, 1)
    tot = sum(buf[i] for i in range(6))
    print("B%d %d->%d %dx%d k%d: %.3f ms; wave-0 cycles over all blocks: %s" % (B, Cin, Cout, H, W, k, e0.elapsed_time(e1),
          ", ".join("%s %.1f%%" % (names[i], 100.0 * buf[i] / tot) for i in range(6))), "| stages", buf[6], "cycles/stage mfma %.0f" % (buf[1] / max(buf[6], 1)))
