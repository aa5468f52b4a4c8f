"""Which kernel should take which convolution?  Records every convolution launch geometry of one B=16 coco-attngan step
(entry point + integer arguments, with its count), then times each one in isolation through the library's dispatch as
configured by the environment of THIS process (MOGAN_WINO, MOGAN_DCONV, ... are read once per process):

    python tools/time_conv_layers.py out.csv          # one row per geometry: count, microseconds, GFLOP

Run it under several environments and compare with tools/time_conv_layers.py --merge a.csv b.csv ...: per geometry the
fastest variant, and the step-time sum per variant / of the per-geometry best."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def merge(paths):
    tabs = []
    for p in paths:
        tabs.append({(r["name"], r["ints"]): r for r in csv.DictReader(open(p))})
    keys = list(tabs[0])
    tot = [0.0] * len(paths)
    best = 0.0
    rows = []
    for k in keys:
        if not all(k in t for t in tabs):
            continue
        us = [float(t[k]["us"]) for t in tabs]
        n = int(tabs[0][k]["count"])
        for i, u in enumerate(us):
            tot[i] += n * u
        best += n * min(us)
        rows.append((n * (us[0] - min(us)), k, n, us))
    rows.sort(reverse=True)
    print("per-step sums (ms): " + "  ".join("%s %.2f" % (os.path.basename(p), t / 1e3) for p, t in zip(paths, tot))
          + "  | best-of %.2f" % (best / 1e3))
    for gain, k, n, us in rows[:40]:
        print("%-22s %-44s x%2d  " % (k[0][6:], k[1], n) + "  ".join("%7.1f" % u for u in us) + "   gain vs first %6.1f us/step" % gain)


def main():
    if sys.argv[1] == "--merge":
        return merge(sys.argv[2:])
    import torch
    import test_fullwidth_parity_gpu as t
    from mogan_amd.attngan import inception
    from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
    from mogan_amd.hip import ops
    set_coco_train_defaults()
    cfg.TRAIN.GENERATOR_LR = cfg.TRAIN.DISCRIMINATOR_LR = 2e-4
    cfg.STN_ALIGN_CORNERS, cfg.ATT_MASK_MODE, cfg.ADAM_EPS_MODE = False, 0, 0
    eng, bt = t._bench_engine(16)
    eng.multi_stream, eng.graph_encoder = False, False
    inception.FAST_TRUNK = False
    counts, orig = {}, ops.call

    def spy(name, *args):
        if name in t._GEMM_ENTRIES and name != "mogan_bmm":
            lo, hi = t._INT_ARGS[name]
            key = (name, tuple(int(a) for a in args[lo:hi]))
            counts[key] = counts.get(key, 0) + 1
        return orig(name, *args)

    ops.call = spy
    eng.step(dict(bt))
    torch.cuda.synchronize()
    ops.call = orig
    del eng
    torch.cuda.empty_cache()
    gen = torch.Generator(device="cuda").manual_seed(7)
    out = []
    for (name, ints), n in counts.items():
        fn = lambda: t._replay(name, ints, gen)
        fn(); fn()
        torch.cuda.synchronize()
        # the replay allocates and fills its operands every call: time only the library call via events around a batch of
        # replays would include the fills -- so take the minimum over replays of (replay - fill-only) ... simpler: hook call
        times = []
        real = ops.call

        def timed(nm, *a):
            if nm == name:
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record(); r = real(nm, *a); e1.record()
                times.append((e0, e1))
                return r
            return real(nm, *a)

        ops.call = timed
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ops.call = real
        us = min(a.elapsed_time(b) for a, b in times) * 1e3 if times else float("nan")
        if name.startswith("mogan_upconv3x3"):
            B, Cin, Hs, Ws, Cout = ints[:5]
            gf = 2.0 * B * Cout * (2 * Hs) * (2 * Ws) * Cin * 9 / 1e9
        else:
            B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw = ints[:10]
            up = ints[10] if name != "mogan_conv2d_affine_fwd" else 0
            OH, OW = ops.conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, up)
            gf = 2.0 * B * Cout * OH * OW * Cin * KH * KW / 1e9
        out.append((name, " ".join(str(i) for i in ints), n, "%.1f" % us, "%.3f" % gf))
    with open(sys.argv[1], "w") as f:
        w = csv.writer(f)
        w.writerow(["name", "ints", "count", "us", "gflop"])
        w.writerows(out)
    tot = sum(int(r[2]) * float(r[3]) for r in out)
    print("%d geometries, %.2f ms per step in isolation" % (len(out), tot / 1e3))


if __name__ == "__main__":
    main()
