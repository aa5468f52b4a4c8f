"""GPU timeline analysis of the multi-stream step from a rocprofv3 kernel trace:
   python tools/timeline.py <kernel_trace.csv> [steps_to_skip_at_start_fraction]
Reports, over the steady-state part of the trace: busy fraction (>= 1 kernel running), mean concurrency, and the kernels with
the most EXCLUSIVE time (running alone) -- the critical path candidates."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r.get("Kernel_Name") or r.get("Name")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = re.sub(r"\(anonymous namespace\)::", "", n); short = re.sub(r"\(.*", "", short)[:60]
    ev.append((s, e, short, r.get("Queue_Id", r.get("Stream_Id", "?"))))
ev.sort()
# steady state = between the ends of two fused-Adam launches of the generator (the last kernel of a step; 4 adam launches per
# step): skip the first 5 steps, take the next 5
adam = [x for x in ev if x[2].startswith("adam_kernel")]
nst = len(adam) // 4
first, last_ = adam[4 * min(5, nst - 6) - 1][1], adam[4 * min(10, nst - 1) - 1][1]
nsteps = min(10, nst - 1) - min(5, nst - 6)
ev = [x for x in ev if x[0] >= first and x[1] <= last_]
print("steps in window: %d, %.2f ms per step under tracing" % (nsteps, (last_ - first) / 1e6 / nsteps))
pts = []
for i, (s, e, n, q) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
active = set(); last = pts[0][0]
busy = 0; conc = 0; excl = collections.Counter(); total = pts[-1][0] - pts[0][0]
hist = collections.Counter()
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        k = len(active)
        hist[min(k, 6)] += dt
        if k >= 1: busy += dt; conc += dt * k
        if k == 1: excl[ev[next(iter(active))][2]] += dt
    if d == 1: active.add(i)
    else: active.discard(i)
    last = t
print("window %.1f ms, %d kernels; busy %.1f %%, mean concurrency while busy %.2f" % (total / 1e6, len(ev), 100.0 * busy / total, conc / max(1, busy)))
print("time by number of concurrently running kernels:", {k: "%.1f%%" % (100.0 * v / total) for k, v in sorted(hist.items())})
print("exclusive (running alone) time by kernel, % of window:")
for n, v in excl.most_common(25):
    print("  %5.1f %%  %s" % (100.0 * v / total, n))
