"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (template arguments kept)."""
import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    m = re.search(r"(dconv_\w+|gemm_kernel|sc_\w+|wino\w+)(<[^>]*>)?", name)
    if not m:
        continue
    k = m.group(0)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, v in sorted(agg.items()):
    print(k, {a: "%.3g" % b for a, b in v.items()})
