#!/bin/bash
# per-kernel launch counts: branch graphs vs eager branches
R=/root/repo; O=$R/gpurun_out/r8u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B2="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
for v in default nobg; do
  E="MOGAN_FAST_INIT=1"; if [ "$v" = nobg ]; then E="$E MOGAN_BRANCH_GRAPHS=0"; fi
  rm -rf /tmp/kc; env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o kc -- $B2 > /tmp/kc.log 2>&1
  cp /tmp/kc/kc_kernel_stats.csv $O/stats_$v.csv
done
