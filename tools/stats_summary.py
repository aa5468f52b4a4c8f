"""Per-step summary of a rocprofv3 kernel-stats csv: python tools/stats_summary.py file.csv [steps=13]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
skip = ('rocsolver', 'rocblas', 'Cijk', 'larf', 'trmm')
mf = ('pgemm_kernel', 'pgemm_group_kernel', 'gemm_kernel', 'gemm_group_kernel', 'dconv_fwd', 'dconv2_fwd', 'dconv_wgrad', 'wino3_fwd', 'wino5_fwd', 'wino_wgrad_kernel', 'stem_k4s2')
out = []
for r in rows:
    n = r['Name']
    if any(s in n for s in skip):
        continue
    short = re.sub(r'\(anonymous namespace\)::', '', n)
    short = re.sub(r'\(.*', '', short)[:80]
    out.append((float(r['TotalDurationNs']) / 1e6 / steps, int(r['Calls']) / steps, short, any(m in n for m in mf)))
out.sort(reverse=True)
for tag, sel in (("MFMA", True), ("other", False)):
    L = [o for o in out if o[3] == sel]
    print("%s: %.2f ms/step, %.0f launches/step" % (tag, sum(o[0] for o in L), sum(o[1] for o in L)))
    for ms, calls, short, _ in L[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
        print("  %7.3f ms %7.1f calls %7.1f us  %s" % (ms, calls, ms / calls * 1e3 if calls else 0, short))
