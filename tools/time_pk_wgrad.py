"""Deep discriminator weight gradients: gemm_kernel (implicit GEMM, mode 2) vs the packed-operand path (two packs + pgemm_kernel).
python tools/time_pk_wgrad.py [cfg] [split]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg  # noqa: E402

load_pkg()
from mogan_amd.hip import lib, ops  # noqa: E402

LAYERS = [  # B, Cin, H, Cout, k, s, p      (B = 32: real and fake parts merged)
    (16, 1536, 8, 3072, 4, 2, 1), (32, 1536, 8, 3072, 4, 2, 1), (32, 768, 16, 1536, 4, 2, 1), (32, 3072, 4, 1536, 3, 1, 1),
    (31, 1024, 4, 768, 3, 1, 1), (32, 384, 16, 384, 4, 2, 1), (32, 1536, 4, 768, 3, 1, 1), (32, 768, 8, 1536, 4, 2, 1),
    (32, 384, 16, 768, 4, 2, 1), (32, 384, 8, 768, 4, 2, 1), (16, 768, 8, 1536, 4, 2, 1),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
    split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib.load().mogan_gemm_set_split_target(768)
    print("layer                              | old ms TF | pk ms TF | rel err")
    for (B, Cin, H, Cout, k, s, p) in LAYERS:
        x = torch.randn(B, Cin, H, H, device="cuda")
        OH = (H + 2 * p - k) // s + 1
        dy = torch.randn(B, Cout, OH, OH, device="cuda")
        gf = 2.0 * B * OH * OH * Cout * Cin * k * k / 1e9
        shape = (Cout, Cin, k, k)
        acc = torch.zeros(shape, device="cuda")
        ops.PK_WGRAD = False
        d0 = ops.conv2d_wgrad(dy, x, shape, s, p, p, 0)
        t0 = timeit(lambda: ops.conv2d_wgrad(dy, x, shape, s, p, p, 0, out=acc, accumulate=True))
        ops.PK_WGRAD = True
        ops.pk_debug_force(0, cfg, split)
        d1 = ops.conv2d_wgrad(dy, x, shape, s, p, p, 0)
        t1 = timeit(lambda: ops.conv2d_wgrad(dy, x, shape, s, p, p, 0, out=acc, accumulate=True))
        err = float((d1 - d0).norm() / d0.norm())
        print("%-34s | %6.3f %5.0f | %6.3f %5.0f | %.1e" % ((B, Cin, H, Cout, k, s, p), t0, gf / t0, t1, gf / t1, err))


if __name__ == "__main__":
    main()
