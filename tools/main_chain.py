"""Kernels of the MAIN stream (the generator's chain: the stream that runs wino5_fwd_kernel) in one steady-state step of a rocprofv3
kernel trace, by name: launches and time, split at the longest idle gap (the window in which the stream waits for the side branches)
into the forward and the backward part.  python tools/main_chain.py <kernel_trace.csv> [step]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r["Queue_Id"])) for r in rows)
adam = [x for x in ev if "adam_kernel" in x[2]]
ends = [adam[i][1] for i in range(3, len(adam), 4)]
t0, t1 = ends[which - 1], ends[which]
win = [x for x in ev if t0 <= x[0] < t1]
main = collections.Counter(x[3] for x in win if "wino5_fwd" in x[2]).most_common(1)[0][0]     # (round 6: wino5_fwd_kernel; wino3 is the encoder stem's)
ms = [x for x in win if x[3] == main]
gap, cut = max((ms[i + 1][0] - ms[i][1], i) for i in range(len(ms) - 1))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:70]


print("step %d: %.2f ms; main stream %s: %d kernels, longest idle gap %.2f ms" % (which, (t1 - t0) / 1e6, main, len(ms), gap / 1e6))
for name, part in (("forward part", ms[:cut + 1]), ("backward part", ms[cut + 1:])):
    cnt, tim = collections.Counter(), collections.Counter()
    for s, e, n, _ in part:
        cnt[short(n)] += 1
        tim[short(n)] += (e - s) / 1e3
    span = (part[-1][1] - part[0][0]) / 1e6
    print("%s: %d kernels, %.2f ms of kernels in a span of %.2f ms" % (name, len(part), sum(tim.values()) / 1e3, span))
    for n, c in sorted(cnt.items(), key=lambda kv: -tim[kv[0]])[:28]:
        print("   %4d %8.1f us  %s" % (c, tim[n], n))
    small = sum(c for n, c in cnt.items() if tim[n] / c < 12.0)
    print("   launches shorter than 12 us on average: %d (%.2f ms)" % (small, sum(tim[n] for n, c in cnt.items() if tim[n] / c < 12.0) / 1e3))
