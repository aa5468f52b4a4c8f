"""Micro-benchmark of the MFMA GEMM kernel: plain bmm (no gather) and the dominant conv shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
dev = "cuda"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg in (0, 1, 4):
    lib.load().mogan_gemm_debug_force(cfg, 1)
    for (M, N, K) in ((4096, 4096, 4096), (96 if cfg == 1 else 128, 262144, 864)):
        a = torch.randn(1, M, K, device=dev); b = torch.randn(1, K, N, device=dev); c = torch.empty(1, M, N, device=dev)
        ms = timeit(lambda: ops.bmm_raw(a, b, c))
        print("bmm cfg%d %dx%dx%d NN: %.3f ms %.1f TF" % (cfg, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
        bt = torch.randn(1, N, K, device=dev)
        ms = timeit(lambda: ops.bmm_raw(a, bt.transpose(1, 2), c))
        print("bmm cfg%d %dx%dx%d NT: %.3f ms %.1f TF" % (cfg, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
lib.load().mogan_gemm_debug_force(-1, 0)
for (B, Cin, H, Cout, k, s, up) in ((16, 96, 128, 192, 3, 1, 0), (16, 96, 128, 96, 3, 1, 1), (16, 192, 64, 384, 4, 2, 0), (16, 1536, 8, 3072, 4, 2, 0)):
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev)
    y = ops.conv2d_forward(x, w, s, 1, 1, up)
    fl = 2.0 * y.numel() * Cin * k * k
    ms = timeit(lambda: ops.conv2d_forward(x, w, s, 1, 1, up))
    print("conv fwd %s: %.3f ms %.1f TF" % ((B, Cin, H, Cout, k, s, up), ms, fl / ms / 1e9))
    dy = torch.randn_like(y)
    ms = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, up))
    print("conv dgrad: %.3f ms %.1f TF" % (ms, fl / ms / 1e9))
    ms = timeit(lambda: ops.conv2d_wgrad(dy, x, w.shape, s, 1, 1, up))
    print("conv wgrad: %.3f ms %.1f TF" % (ms, fl / ms / 1e9))
