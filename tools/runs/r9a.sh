#!/bin/bash
# launch modes again, now that the host side is light (text encoder = one launch): graphs vs eager branches / generator
R=/root/repo; O=$R/gpurun_out/r9a; mkdir -p $O; cd $R
for i in 1 2; do for v in "X=1" "MOGAN_BRANCH_GRAPHS=0" "MOGAN_G_GRAPHS=0" "MOGAN_BRANCH_GRAPHS=0 MOGAN_G_GRAPHS=0" "MOGAN_GRAPH_ENCODER=0"; do
    echo -n "$v  " >> $O/ab.txt
    env $v timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms host %.1f' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))" >> $O/ab.txt
done; done
