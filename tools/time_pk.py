"""Deep discriminator layers: implicit-GEMM kernels (gemm_kernel) vs the packed-weight path (pgemm_kernel), isolated, B = 16.
python tools/time_pk.py [cfg] [split]      (cfg / split: forced tile shape / K-split of the packed path, default heuristic)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg  # noqa: E402

load_pkg()
from mogan_amd.hip import lib, ops  # noqa: E402

LAYERS = [  # B, Cin, H, Cout, k, s, p
    (16, 1536, 8, 3072, 4, 2, 1), (16, 768, 16, 1536, 4, 2, 1), (16, 3072, 4, 1536, 3, 1, 1), (16, 1024, 4, 768, 3, 1, 1),
    (16, 384, 16, 384, 4, 2, 1), (16, 1536, 4, 768, 3, 1, 1), (16, 768, 8, 1536, 4, 2, 1), (16, 384, 16, 768, 4, 2, 1),
    (16, 384, 8, 768, 4, 2, 1), (15, 1024, 4, 768, 3, 1, 1),
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
    lib.load().mogan_pk_debug_force(0, cfg, split)
    lib.load().mogan_gemm_set_split_target(768)
    print("layer                              | fwd old ms TF | fwd pk ms TF | dgrad old ms TF | dgrad pk ms TF | pack fwd+dg ms")
    for (B, Cin, H, Cout, k, s, p) in LAYERS:
        x = torch.randn(B, Cin, H, H, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
        OH = (H + 2 * p - k) // s + 1
        dy = torch.randn(B, Cout, OH, OH, device="cuda")
        gf = 2.0 * B * OH * OH * Cout * Cin * k * k / 1e9
        w0 = w.clone()
        t_f0 = timeit(lambda: ops.conv2d_forward(x, w0, s, p, p, 0))
        t_d0 = timeit(lambda: ops.conv2d_dgrad(dy, w0, x.shape, s, p, p, 0))
        pk = ops.attach_packs(w)
        y1 = ops.conv2d_forward(x, w, s, p, p, 0)
        d1 = ops.conv2d_dgrad(dy, w, x.shape, s, p, p, 0)
        y0 = ops.conv2d_forward(x, w0, s, p, p, 0)
        d0 = ops.conv2d_dgrad(dy, w0, x.shape, s, p, p, 0)
        ey = float((y1 - y0).norm() / y0.norm())
        ed = float((d1 - d0).norm() / d0.norm())
        t_f1 = timeit(lambda: ops.conv2d_forward(x, w, s, p, p, 0))
        t_d1 = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p, p, 0))
        t_p = timeit(lambda: pk.repack())
        print("%-34s | %6.3f %6.1f | %6.3f %6.1f | %6.3f %6.1f | %6.3f %6.1f | %6.3f   (rel diff %.1e %.1e, slots %s)" % (
            (B, Cin, H, Cout, k, s), t_f0, gf / t_f0, t_f1, gf / t_f1, t_d0, gf / t_d0, t_d1, gf / t_d1, t_p, ey, ed,
            sorted(pk.slots)))


if __name__ == "__main__":
    main()
