"""Accuracy of the loaded library's MFMA kernels against fp64 on the bench geometries and on adversarial data.

Run it twice -- default library (split-bf16 form, csrc/mogan_mma.h) and MOGAN_LIB=multiple-objects-gan_amd/libmogan_hip_f32.so (native
fp32-MFMA form) -- and compare the columns: both carry fp32 rounding only if the numbers agree to within a small factor.
    python tools/diag_x6_precision.py            # implicit-GEMM + direct kernels (MOGAN_WINO=0 set here)
    python tools/diag_x6_precision.py wino       # default dispatch (Winograd kernels where eligible)
Data sets: "normal" = N(0,1) activations, He-scaled weights; "wide" = every value multiplied by 2^U(-20,20) (products of
very different magnitude in one sum); "cancel" = x and -x interleaved along the reduction with 1e-4 relative noise (the
sum is ~1e-4 of the terms: error relative to sum |a||b| is what fp32 can hold)."""
import os
import sys

if "wino" not in sys.argv[1:]:
    os.environ["MOGAN_WINO"] = "0"
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg  # noqa: E402

load_pkg()
from mogan_amd.hip import ops  # noqa: E402

LAYERS = [(96, 128, 192, 4, 2, 1, 8), (384, 32, 768, 4, 2, 1, 16), (768, 16, 1536, 4, 2, 1, 16), (1536, 8, 3072, 4, 2, 1, 16),
          (3072, 4, 1536, 3, 1, 1, 16), (96, 64, 192, 3, 1, 1, 8), (96, 128, 96, 3, 1, 1, 4), (192, 35, 64, 1, 1, 0, 16),
          (768, 17, 192, 1, 1, 0, 16), (288, 35, 384, 3, 2, 0, 16)]


def err(got, want, scale):
    """max |got - want| / scale and rel-L2; scale = the fp64 value of sum |a||b| (what bounds fp32 rounding)"""
    d = (got.double().cpu() - want).abs()
    return float((d / scale.clamp_min(1e-300)).max()), float(d.norm() / want.norm().clamp_min(1e-300))


def make(kind, shape, g):
    x = torch.randn(shape, generator=g)
    if kind == "wide":
        x = x * torch.exp2(torch.empty(shape).uniform_(-20, 20, generator=g))
    return x


def main():
    print("library:", os.environ.get("MOGAN_LIB", "default (in-tree)"))
    for kind in ("normal", "wide", "cancel"):
        print("== data:", kind)
        for Cin, H, Cout, k, s, p, B in LAYERS:
            g = torch.Generator().manual_seed(Cin * 7 + H)
            x = make(kind, (B, Cin, H, H), g)
            w = make(kind, (Cout, Cin, k, k), g) * (1.0 / (Cin * k * k)) ** 0.5
            if kind == "cancel":                        # channel 2i+1 = -(channel 2i) (1 + 1e-4 noise), same weights
                x[:, 1::2] = -x[:, 0::2][:, :x[:, 1::2].shape[1]] * (1 + 1e-4 * torch.randn(x[:, 1::2].shape, generator=g))
                w[:, 1::2] = w[:, 0::2][:, :w[:, 1::2].shape[1]]
            xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
            yd = F.conv2d(xd, wd, None, s, p)
            dy = make("normal" if kind == "cancel" else kind, yd.shape, g)
            yd.backward(dy.double())
            ya = F.conv2d(x.double().abs(), w.double().abs(), None, s, p)
            xa = x.double().abs().requires_grad_(True)
            wa = w.double().abs().requires_grad_(True)
            F.conv2d(xa, wa, None, s, p).backward(dy.double().abs())
            xg, wg, dyg = x.cuda(), w.cuda(), dy.cuda()
            y = ops.conv2d_forward(xg, wg, s, p, p, 0)
            dx = ops.conv2d_dgrad(dyg, wg, xg.shape, s, p, p, 0)
            dw = ops.conv2d_wgrad(dyg, xg, wg.shape, s, p, p, 0)
            torch.cuda.synchronize()
            e = [err(y, yd.detach(), ya), err(dx, xd.grad, xa.grad), err(dw, wd.grad, wa.grad)]
            print("B=%2d %4d->%4d %3dx%-3d k%d s%d | fwd max/scale %.2e relL2 %.2e | dgrad %.2e %.2e | wgrad %.2e %.2e"
                  % (B, Cin, Cout, H, H, k, s, e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1]), flush=True)
        for (bz, M, N, K) in ((1, 1024, 1024, 4096), (16, 256, 289, 768), (4, 100, 4096, 96)):
            g = torch.Generator().manual_seed(M + N + K)
            a = make(kind, (bz, M, K), g)
            b = make(kind, (bz, K, N), g)
            if kind == "cancel":
                a[:, :, 1::2] = -a[:, :, 0::2] * (1 + 1e-4 * torch.randn(a[:, :, 1::2].shape, generator=g))
                b[:, 1::2, :] = b[:, 0::2, :]
            want = torch.bmm(a.double(), b.double())
            scale = torch.bmm(a.double().abs(), b.double().abs())
            got = ops.bmm(a.cuda(), b.cuda())
            torch.cuda.synchronize()
            e = err(got, want, scale)
            print("bmm %2d x (%4d x %4d x %4d)         | max/scale %.2e relL2 %.2e" % (bz, M, N, K, e[0], e[1]), flush=True)


if __name__ == "__main__":
    main()
