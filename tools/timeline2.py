"""Per-queue Gantt of one steady-state step from a rocprofv3 kernel trace: python tools/timeline2.py <kernel_trace.csv> [step]
One row per hardware queue, one character per 0.5 ms bin = the kernel family with the most time in that bin."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 7
FAM = [("pgemm", "P"), ("gemm_kernel", "G"), ("gemm_group", "G"), ("dconv_fwd", "D"), ("dconv_wgrad", "d"), ("wino_fwd", "W"), ("wino_wgrad", "w"),
       ("wpack", "k"), ("apack", "k"), ("deep_tail", "t"), ("bn_", "b"), ("splitk", "r"), ("dconv_reduce", "r"), ("adam", "A"), ("damsm", "m"),
       ("attn", "a"), ("stn", "s"), ("act_kernel", "e"), ("glu", "e"), ("sc_", "c"), ("Cijk", "L"), ("elementwise", "x"), ("copy", "x"), ("Cat", "x")]
def fam(n):
    for k, c in FAM:
        if k in n: return c
    return "o"
ev = []
for r in rows:
    n = r.get("Kernel_Name") or r.get("Name")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
ev.sort()
adam = [x for x in ev if "adam_kernel" in x[2]]
ends = [adam[i][1] for i in range(3, len(adam), 4)]          # end of the 4th Adam of every step (the generator's)
t0, t1 = ends[which - 1], ends[which]
win = [x for x in ev if x[1] > t0 and x[0] < t1]
print("step %d: %.2f ms, %d kernels" % (which, (t1 - t0) / 1e6, len(win)))
BIN = 500000
nb = int((t1 - t0) / BIN) + 1
qs = sorted(set(x[3] for x in win))
for q in qs:
    bins = [collections.Counter() for _ in range(nb)]
    busy = 0; cnt = 0
    for s, e, n, qq in win:
        if qq != q: continue
        s2, e2 = max(s, t0), min(e, t1)
        busy += e2 - s2; cnt += 1
        b = int((s2 - t0) / BIN)
        while s2 < e2:
            be = t0 + (b + 1) * BIN
            bins[b][fam(n)] += min(e2, be) - s2
            s2 = be; b += 1
    row = "".join((c.most_common(1)[0][0] if c and sum(c.values()) > 0.25 * BIN else ("." if c else " ")) for c in bins)
    print("q%-3s busy %6.2f ms %5d kernels |%s|" % (q, busy / 1e6, cnt, row))
tot = collections.Counter(); num = collections.Counter()
for s, e, n, q in win:
    tot[fam(n)] += e - s; num[fam(n)] += 1
print("family totals (ms, launches):", {k: (round(v / 1e6, 2), num[k]) for k, v in tot.most_common()})
