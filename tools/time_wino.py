"""Winograd F(2x2,3x3) lab kernel (csrc/mogan_wino.hip) against the direct convolution on the ResBlock shapes."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
L = lib.load()
P, I = ctypes.c_void_p, ctypes.c_int
L.mogan_lab_wino_weights.argtypes = [P, P, I, I, I, P]
L.mogan_lab_wino_fwd.argtypes = [P, P, P, I, I, I, I, I, P]
dev = "cuda"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, Cin, H, Cout) in [(2, 16, 8, 16), (2, 32, 32, 100), (16, 96, 128, 192), (16, 96, 128, 96), (16, 96, 64, 192), (16, 96, 64, 96), (16, 192, 64, 96)]:
    W = max(H, 32)
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    U = torch.zeros(16 * Cin * ((Cout + 95) // 96) * 96, device=dev); y = torch.empty(B, Cout, H, W, device=dev)
    st = lib.stream_ptr()
    def wino():
        assert L.mogan_lab_wino_weights(w.data_ptr(), U.data_ptr(), Cout, Cin, 0, st) == 0
        assert L.mogan_lab_wino_fwd(x.data_ptr(), U.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, st) == 0
    wino()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    yd = ops.conv2d_forward(x, w, 1, 1, 1, 0)
    errd = (yd.double() - ref).abs().max().item() / ref.abs().max().item()
    # dgrad through the same kernel: flip
    dy = torch.randn(B, Cout, H, W, device=dev)
    dx = torch.empty(B, Cin, H, W, device=dev); U2 = torch.zeros(16 * Cout * ((Cin + 95) // 96) * 96, device=dev)
    L.mogan_lab_wino_weights(w.data_ptr(), U2.data_ptr(), Cout, Cin, 1, st)
    rc = L.mogan_lab_wino_fwd(dy.data_ptr(), U2.data_ptr(), dx.data_ptr(), B, Cout, H, W, Cin, st)
    refd = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    errg = (dx.double() - refd).abs().max().item() / refd.abs().max().item() if rc == 0 else float("nan")
    tw, td = t(wino), t(lambda: ops.conv2d_forward(x, w, 1, 1, 1, 0))
    gf = 2.0 * B * H * W * Cout * Cin * 9 / 1e9
    print("B%d %d->%d %dx%d: wino %.3f ms (%.0f TF-equiv, rel err %.1e, dgrad err %.1e) | direct %.3f ms (%.0f TF, err %.1e)"
          % (B, Cin, Cout, H, W, tw, gf / tw, err, errg, td, gf / td, errd), flush=True)
