#!/bin/bash
# step A/B: wino_wgrad2_kernel (block counts) against wino_wgrad_kernel
R=/root/repo; O=$R/gpurun_out/r8i; mkdir -p $O; cd $R
for i in 1 2; do for v in "MOGAN_WG2=0" "MOGAN_WG2_BLOCKS=512" "MOGAN_WG2_BLOCKS=384" "MOGAN_WG2_BLOCKS=448"; do
    echo -n "$v  " >> $O/ab.txt
    env $v MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
done; done
