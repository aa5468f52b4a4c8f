#!/bin/bash
# blocks per launch of wino_wgrad_kernel (192 since round 3) on the round-6 tree (lab libraries)
R=/root/repo; O=$R/gpurun_out/r8z; mkdir -p $O; cd $R
for i in 1 2; do for v in product wgb_160 wgb_224 wgb_256; do
    if [ $v = product ]; then unset MOGAN_LIB; else export MOGAN_LIB=$R/tools/lab/libmogan_$v.so; fi
    echo -n "$v  " >> $O/ab.txt
    MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d['chain_ms'].get('G backward'))" >> $O/ab.txt
done; done
