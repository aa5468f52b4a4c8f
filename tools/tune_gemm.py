"""Is run_gemm's tile / split-K choice the best one?  For every distinct conv shape the generic implicit-GEMM kernel
sees in a train step (gpurun_out/layers.csv from tools/profile_layers.py) time the default choice against every forced
(tile config, split) pair.  Output: one line per shape, worst offenders first."""
import csv, math, os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
dev = "cuda"
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "layers.csv"))))
geo = {}
for r in rows:
    if r["mode"] == "0" and int(r["Cin"]) > 0:
        B, H, W, KH, KW, s = (int(r[k]) for k in ("B", "H", "W", "KH", "KW", "stride"))
        ohw = int(r["N"]) // B
        OH = OW = int(round(math.sqrt(ohw)))
        geo[(r["Cin"], r["Cout"], r["H"], r["W"], r["KH"], r["KW"], r["stride"], r["up"])] = \
            (max(0, ((OH - 1) * s + KH - H + 1) // 2), max(0, ((OW - 1) * s + KW - W + 1) // 2))
shapes = collections.OrderedDict()
for r in rows:
    if r["mode"] in "012" and int(r["Cin"]) > 0:
        k = tuple(r[x] for x in ("mode", "B", "Cin", "Cout", "H", "W", "KH", "KW", "stride", "up"))
        a = shapes.setdefault(k, [0, r["cfg"], r["nsplit"], r["M"], r["N"], r["K"], r["nz"]]); a[0] += 1

def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

L = lib.load()
out = []
allrows = []
for k, (cnt, cfg0, ns0, gM, gN, gK, gnz) in shapes.items():
    mode, B, Cin, Cout, H, W, KH, KW, s, up = (int(v) for v in k)
    ph, pw = geo.get(tuple(str(v) for v in (Cin, Cout, H, W, KH, KW, s, up)), (KH // 2, KW // 2))
    Hs, Ws = H >> up, W >> up
    x = torch.randn(B, Cin, Hs, Ws, device=dev); w = torch.randn(Cout, Cin, KH, KW, device=dev) * 0.05
    L.mogan_gemm_debug_force(-1, 0)
    y = ops.conv2d_forward(x, w, s, ph, pw, up); dy = torch.randn_like(y); g = torch.zeros_like(w)
    fn = {0: lambda: ops.conv2d_forward(x, w, s, ph, pw, up),
          1: lambda: ops.conv2d_dgrad(dy, w, x.shape, s, ph, pw, up),
          2: lambda: ops.conv2d_wgrad(dy, x, w.shape, s, ph, pw, up, out=g, accumulate=True)}[mode]
    L.mogan_gemm_debug_force(int(cfg0), int(ns0))          # what the step used (forced: bypasses the direct kernels)
    base = t(fn)
    best = (base, int(cfg0), int(ns0))
    for c in range(7):
        for sp in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32, 48):
            L.mogan_gemm_debug_force(c, sp)
            try:
                v = t(fn, 4)
            except Exception:
                continue
            if v < best[0]: best = (v, c, sp)
            allrows.append(list(k) + [ph, pw, gM, gN, gK, gnz, cnt, cfg0, ns0, c, sp, "%.2f" % v])
    L.mogan_gemm_debug_force(-1, 0)
    out.append((cnt * (base - best[0]), k, cnt, base, cfg0, ns0, best))
with open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "tune_gemm_all.csv"), "w") as f:
    w_ = csv.writer(f); w_.writerow(["mode", "B", "Cin", "Cout", "H", "W", "KH", "KW", "stride", "up", "ph", "pw", "M", "N", "K", "nz",
                                     "count", "cfg0", "split0", "cfg", "split", "us"])
    w_.writerows(allrows)
out.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in out); tb = sum(r[2] * r[3] for r in out)
print("total default %.2f ms/step, best-of-all %.2f ms/step" % (tb / 1e3, (tb - tot) / 1e3))
for gain, k, cnt, base, cfg0, ns0, best in out[:60]:
    print("%-44s x%2d  default cfg %s split %2s %7.1f us | best cfg %d split %2d %7.1f us | gain %6.1f us/step"
          % (" ".join(k), cnt, cfg0, ns0, base, best[1], best[2], best[0], gain))
