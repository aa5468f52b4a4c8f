#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8x; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "text_encoder_as_one" 2>&1 | tail -12 > $O/tests.txt
for b in 4 8; do for v in on off; do
    echo -n "B=$b fused=$v  " >> $O/ab.txt
    if [ $v = off ]; then A="mogan_amd.attngan.model:RNN_ENCODER.FUSED=False"; else A="mogan_amd.attngan.model:RNN_ENCODER.FUSED=True"; fi
    timeout 600 python tools/ab_attr.py $A -- bench.py --batch $b --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms host %.1f' % (d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step')))" >> $O/ab.txt
done; done
