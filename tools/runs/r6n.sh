#!/bin/bash
cd /root/repo; O=gpurun_out/r6n; mkdir -p $O
timeout 300 python tools/time_wino_prep.py 2>&1 | grep -v amdgpu.ids > $O/prep.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python /root/repo/tools/time_wino_prep.py > /dev/null 2>&1
python - <<'PY' >> /root/repo/gpurun_out/r6n/prep.txt
import csv,glob
f=glob.glob('/tmp/pp/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
