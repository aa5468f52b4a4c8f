#!/usr/bin/env python
"""HBM-side traffic per launch of the MFMA kernels from two rocprofv3 PMC passes (the guide's recipe: FETCH_SIZE and
WRITE_SIZE in SEPARATE passes, --kernel-trace only).

  cd /tmp && export TMPDIR=/tmp
  E="MOGAN_FAST_INIT=1 MOGAN_STREAMS=0 MOGAN_WGRAD_STREAM=0"      # one stream: a dispatch's counters are its own
  env $E rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
  env $E rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_traffic.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv profiles/r01_pmc_traffic.json

Units: FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
counts a wide (16 B/lane) coalesced read at half its bytes; these kernels mix 16-byte weight/dY loads with dword halo
gathers, so both the raw value and the doubled value (an upper bound) are stored and bench.py reports
traffic = 2*fetch + write.  Infinity-Cache hits are included in the counters."""
import csv
import json
import re
import sys
from collections import defaultdict

KEEP = ("gemm_kernel", "pgemm_kernel", "pgemm_group_kernel", "dconv_fwd_kernel", "dconv2_fwd_kernel", "dconv_wgrad_kernel",
        "wino3_fwd_kernel", "wino5_fwd_kernel", "wino_wgrad_kernel")


def short(name):
    if "pgemm_group_kernel" not in name:
        name = name.replace("gemm_group_kernel", "gemm_kernel")      # grouped launches of the same tile kernel
    m = re.search(r"(pgemm_group_kernel|pgemm_kernel|gemm_kernel|dconv2_fwd_kernel|dconv_fwd_kernel|dconv_wgrad_kernel|wino3_fwd_kernel)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2))
    for k in ("wino3_fwd_kernel", "wino5_fwd_kernel", "wino_wgrad_kernel"):
        if k in name:
            return k + "<>"
    return None


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k:
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


def main():
    f, w, out = sys.argv[1:4]
    fe, wr = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe[k][0] + wr[k][0])):
        n = fe[k][1] or wr[k][1]
        kernels[k] = {"fetch_kb_per_launch": round(fe[k][0] / max(1, fe[k][1]), 1),
                      "fetch_kb_per_launch_x2": round(2 * fe[k][0] / max(1, fe[k][1]), 1),
                      "write_kb_per_launch": round(wr[k][0] / max(1, wr[k][1]), 1), "launches": n}
    json.dump({"note": __doc__, "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
