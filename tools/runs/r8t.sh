#!/bin/bash
# where the ~170 D2D copy launches of a step come from: eager vs graph modes
R=/root/repo; O=$R/gpurun_out/r8t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B2="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
for v in "default" "MOGAN_BRANCH_GRAPHS=0" "MOGAN_G_GRAPHS=0" "MOGAN_GRAPH_ENCODER=0" "MOGAN_BRANCH_GRAPHS=0 MOGAN_G_GRAPHS=0 MOGAN_GRAPH_ENCODER=0"; do
  E="MOGAN_FAST_INIT=1"; if [ "$v" != default ]; then E="$E $v"; fi
  rm -rf /tmp/kc; env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o kc -- $B2 > /tmp/kc.log 2>&1
  echo "== $v" >> $O/copies.txt
  python - >> $O/copies.txt <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/kc/kc_kernel_stats.csv')))
tot=sum(int(r['Calls']) for r in rows)
for r in rows:
    if 'copyBuffer' in r['Name'] or 'fillBuffer' in r['Name']:
        print(r['Name'][:40], 'calls/step %.1f' % (int(r['Calls'])/13.0), 'us/step %.1f' % (float(r['TotalDurationNs'])/13e3))
print('total launches/step %.1f' % (tot/13.0))
PY
done
