"""Per-kernel-family MFMA-busy from one rocprofv3 PMC pass (tools/pmc_mfma.sh): SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE.
python tools/pmc_mfma.py <counter_collection.csv> <kernel_trace.csv> <out.json>"""
import csv, json, re, sys, collections
cc, kt, out = sys.argv[1:4]
FAM = ("pgemm_group_kernel", "pgemm_kernel", "gemm_group_kernel", "gemm_kernel", "dconv2_fwd_kernel", "dconv_fwd_kernel",
       "dconv_wgrad_kernel", "wino3_fwd_kernel", "wino5_fwd_kernel", "wino_wgrad_kernel")      # (longer names first: "gemm_kernel" is a substring of three)
def fam(name):
    for f in FAM:
        if f in name:
            return "gemm_kernel" if f == "gemm_group_kernel" else f
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(cc)):
    f = fam(r["Kernel_Name"])
    if f:
        acc[f][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[f] += 1
res = {}
for f, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    res[f] = {"launches": n[f], "counters": {k: v for k, v in c.items()},
              # MFMA busy cycles are summed over the 1024 SIMDs (4 per CU x 256 CUs); GRBM_GUI_ACTIVE is accumulated over the 8
              # XCDs (each has its own GRBM), so a kernel's wall cycles are GUI_ACTIVE / 8.  Calibration on this very pass:
              # gemm_kernel 0.43 and dconv_fwd_kernel 0.79 against 0.44 / 0.74 from flops / time / 157.3 TF.
              "mfma_busy_per_simd_cycle": mf / (gui / 8.0 * 1024.0) if gui else None,
              "wait_inst_any_per_wave_cycle": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
              "wait_any_per_wave_cycle": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}
json.dump({"note": __doc__, "families": res}, open(out, "w"), indent=1)
for f, v in sorted(res.items(), key=lambda kv: -kv[1]["counters"].get("GRBM_GUI_ACTIVE", 0)):
    print("%-22s launches %4d  MFMA busy / (GUI_ACTIVE/8 x 1024 SIMDs) = %s   WAIT_INST_ANY/WAVE_CYCLES %s  WAIT_ANY/WAVE_CYCLES %s"
          % (f, v["launches"], "%.3f" % v["mfma_busy_per_simd_cycle"] if v["mfma_busy_per_simd_cycle"] is not None else "-",
             "%.2f" % v["wait_inst_any_per_wave_cycle"] if v["wait_inst_any_per_wave_cycle"] is not None else "-",
             "%.2f" % v["wait_any_per_wave_cycle"] if v["wait_any_per_wave_cycle"] is not None else "-"))
