"""Deep D layers: forward / data gradient time of the 128x128 implicit GEMM under forced split-K factors (isolated)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
L = lib.load()
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Cin, Cout, H, k, s) in [(1536, 3072, 8, 4, 2), (768, 1536, 16, 4, 2), (3072, 1536, 4, 3, 1), (384, 768, 16, 4, 2)]:
    B = 16
    x = torch.randn(B, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
    y = ops.conv2d_forward(x, w, s, 1, 1, 0); dy = torch.randn_like(y)
    gf = 2.0 * Cout * y.shape[0] * y.shape[2] * y.shape[3] * Cin * k * k / 1e9
    out = []
    for cfg in (0, 5):
        for sp in (2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 14, 16, 20, 21, 24, 32):
            L.mogan_gemm_debug_force(cfg, sp)
            tf = t(lambda: ops.conv2d_forward(x, w, s, 1, 1, 0)); td = t(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, 0))
            out.append((cfg, sp, gf / tf, gf / td))
    L.mogan_gemm_debug_force(-1, 0)
    tf = t(lambda: ops.conv2d_forward(x, w, s, 1, 1, 0)); td = t(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, 0))
    print("%4d->%4d %2dx%-2d k%d: default fwd %.1f dgrad %.1f TF" % (Cin, Cout, H, H, k, gf / tf, gf / td))
    bf = max(out, key=lambda o: o[2]); bd = max(out, key=lambda o: o[3])
    print("   best fwd cfg %d split %d: %.1f TF; best dgrad cfg %d split %d: %.1f TF" % (bf[0], bf[1], bf[2], bd[0], bd[1], bd[3]))
    print("   fwd by split (cfg0):", " ".join("%d:%.0f" % (o[1], o[2]) for o in out if o[0] == 0))
    print("   dgrad by split (cfg0):", " ".join("%d:%.0f" % (o[1], o[3]) for o in out if o[0] == 0))
