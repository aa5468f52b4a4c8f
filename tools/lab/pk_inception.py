"""Feasibility probe: the Inception trunk's convolution shapes (B = 16) on the packed-weight kernel vs the implicit-GEMM kernel.
Per-launch times of the MFMA kernels come from the library's own launch profile (mogan_prof_*), so the activation pack and the
split-K reduction are not in them.   python tools/lab/pk_inception.py [cfg] [split] [target]"""
import csv
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogan_loader  # noqa: E402

mogan_loader.load()
from mogan_amd.hip import lib, ops  # noqa: E402

# Cin, H, W, Cout, KH, KW, stride, ph, pw
SHAPES = [
    (192, 35, 35, 208, 1, 1, 1, 0, 0), (288, 35, 35, 240, 1, 1, 1, 0, 0), (64, 35, 35, 64, 5, 5, 1, 2, 2),
    (64, 35, 35, 96, 3, 3, 1, 1, 1), (96, 35, 35, 96, 3, 3, 1, 1, 1), (288, 35, 35, 384, 3, 3, 2, 0, 0),
    (768, 17, 17, 640, 1, 1, 1, 0, 0), (768, 17, 17, 768, 1, 1, 1, 0, 0), (128, 17, 17, 128, 1, 7, 1, 0, 3),
    (160, 17, 17, 160, 7, 1, 1, 3, 0), (192, 17, 17, 192, 1, 7, 1, 0, 3), (192, 17, 17, 320, 3, 3, 2, 0, 0),
    (1280, 8, 8, 1344, 1, 1, 1, 0, 0), (2048, 8, 8, 1344, 1, 1, 1, 0, 0), (384, 8, 8, 384, 1, 3, 1, 0, 1),
    (448, 8, 8, 384, 3, 3, 1, 1, 1),
]


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
    split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    target = int(sys.argv[3]) if len(sys.argv) > 3 else 384
    L = lib.load()
    L.mogan_pk_debug_force(1, cfg, split)
    L.mogan_gemm_set_split_target(target)
    B = 16
    out = os.path.join(ROOT, "gpurun_out", "pk_inc_layers.csv")
    rows = []
    for (Cin, H, W, Cout, KH, KW, s, ph, pw) in SHAPES:
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.02
        OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
        gf = 2.0 * B * OH * OW * Cout * Cin * KH * KW / 1e9
        y0 = torch.empty(B, Cout, OH, OW, device="cuda")
        y1 = torch.empty_like(y0)
        nb = L.mogan_pk_weight_bytes(Cout, Cin, KH, KW, s, 0)
        wpk = torch.empty(nb, dtype=torch.uint8, device="cuda")
        st = lib.stream_ptr()
        lib.call("mogan_pk_weight_pack", w.data_ptr(), wpk.data_ptr(), Cout, Cin, KH, KW, s, ph, pw, 0, st)
        wsp, wsn = lib.workspace(x.device)
        res = {}
        for name in ("old", "pk"):
            def run():
                if name == "old":
                    lib.call("mogan_conv2d_fwd", x.data_ptr(), w.data_ptr(), y0.data_ptr(), B, Cin, H, W, Cout, KH, KW, s, ph, pw, 0,
                             wsp, wsn, st)
                else:
                    lib.call("mogan_conv2d_fwd_pk", x.data_ptr(), wpk.data_ptr(), y1.data_ptr(), B, Cin, H, W, Cout, KH, KW, s, ph, pw,
                             wsp, wsn, st)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            lib.call("mogan_prof_enable", 1)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            lib.call("mogan_prof_dump", out.encode())
            lib.call("mogan_prof_enable", 0)
            r = list(csv.DictReader(open(out)))
            ms = sorted(float(q["ms"]) for q in r)[len(r) // 2]
            res[name] = (ms, r[0]["cfg"], r[0]["nsplit"], r[0]["mode"])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[name] += (e0.elapsed_time(e1) / 20,)
        err = float((y1 - y0).norm() / y0.norm())
        print("%-40s %6.2f GF | old %6.1f us %5.1f TF (mode %s cfg %s ns %s; all %6.1f us) | pk %6.1f us %5.1f TF (cfg %s ns %s; all %6.1f us) "
              "| rel %.1e" % ((Cin, H, W, Cout, KH, KW, s), gf, res["old"][0] * 1e3, gf / res["old"][0], res["old"][3], res["old"][1],
                              res["old"][2], res["old"][4] * 1e3, res["pk"][0] * 1e3, gf / res["pk"][0], res["pk"][1], res["pk"][2],
                              res["pk"][4] * 1e3, err), flush=True)
        rows.append((gf, res["old"][0], res["pk"][0]))
    g = sum(r[0] for r in rows)
    print("sum: %.1f GF, old %.1f us (%.1f TF), pk %.1f us (%.1f TF)" % (g, sum(r[1] for r in rows) * 1e3, g / sum(r[1] for r in rows),
                                                                          sum(r[2] for r in rows) * 1e3, g / sum(r[2] for r in rows)))


if __name__ == "__main__":
    main()
