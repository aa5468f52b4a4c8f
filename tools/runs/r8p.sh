#!/bin/bash
# split-K block target of pgemm_kernel alone (lab library: -DPK_LAB_TARGET, MOGAN_PK_TARGET_PCT = percent of the stream's target)
R=/root/repo; O=$R/gpurun_out/r8p; mkdir -p $O; cd $R
export MOGAN_LIB=$R/tools/lab/libmogan_pktarget.so
for i in 1 2; do for v in 100 50 150 200 300; do
    echo -n "pct=$v  " >> $O/ab.txt
    MOGAN_PK_TARGET_PCT=$v timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']))" >> $O/ab.txt
done; done
