#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8v; mkdir -p $O; cd $R
python tools/time_text.py 2>&1 | grep "text encoder" > $O/text.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/tools/time_text.py > /tmp/kt.log 2>&1
python - >> $O/text.txt <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/kt/kt_kernel_stats.csv')))
tot=sum(int(r['Calls']) for r in rows); tt=sum(float(r['TotalDurationNs']) for r in rows)
print('launches per call %.1f, kernel time per call %.1f us' % (tot/55.0, tt/55e3))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:12]:
    print('  %-70s %6.1f calls %7.1f us' % (r['Name'][:70], int(r['Calls'])/55.0, float(r['TotalDurationNs'])/55e3))
PY
