"""tools/tune_gemm.py timing files (one per workload) -> multiple-objects-gan_amd/hip/tuned_gemm_gfx950.csv: for every GEMM
where the best (tile config, split) pair beats the heuristic's own pair by >= 3 %, one line `mode,M,N,K,nz,cfg,split`.
Usage: python tools/make_tuned_table.py gpurun_out/tune_*.csv"""
import csv, collections, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "tune_*.csv")))
SPL = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48]
table, total = {}, 0.0
for fn in files:
    T, meta = collections.defaultdict(dict), {}
    for r in csv.DictReader(open(fn)):
        k = tuple(int(r[x]) for x in ("mode", "M", "N", "K", "nz"))
        T[k][(int(r["cfg"]), int(r["split"]))] = min(float(r["us"]), T[k].get((int(r["cfg"]), int(r["split"])), 1e30))
        meta[k] = (int(r["count"]), int(r["cfg0"]), int(r["split0"]))
    gain = 0.0
    for k, t in T.items():
        cnt, c0, ns0 = meta[k]
        base = t.get((c0, min(SPL, key=lambda s: abs(s - ns0))))
        (c, sp), v = min(t.items(), key=lambda kv: kv[1])
        if base is None or v < 0.97 * base:
            table[k] = (c, sp, base or 0.0, v)
            gain += cnt * ((base or v) - v)
    print("%-40s %3d GEMMs, isolated gain %.2f ms/step" % (os.path.basename(fn), len(T), gain / 1e3))
path = os.path.join(ROOT, "multiple-objects-gan_amd", "hip", "tuned_gemm_gfx950.csv")
with open(path, "w") as f:
    f.write("# mode,M,N,K,nz,cfg,split,heuristic_us,tuned_us   (MI355X; tools/tune_gemm.py on the benchmark workloads)\n")
    for k in sorted(table):
        f.write("%d,%d,%d,%d,%d,%d,%d,%.1f,%.1f\n" % (k + table[k]))
print("wrote %s: %d entries" % (path, len(table)))
