#!/bin/bash
# size threshold of the packed-operand weight gradient in the step (lab libraries: dW >= 4M / 2M elements, K >= 512 / 256, against 16M / 512)
R=/root/repo; O=$R/gpurun_out/r8y; mkdir -p $O; cd $R
for i in 1 2; do for v in product pkwg_4_512 pkwg_4_256 pkwg_2_256; do
    if [ $v = product ]; then unset MOGAN_LIB; else export MOGAN_LIB=$R/tools/lab/libmogan_$v.so; fi
    echo -n "$v  " >> $O/ab.txt
    timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']))" >> $O/ab.txt
done; done
