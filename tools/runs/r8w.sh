#!/bin/bash
# the text encoder as one launch: tests, time alone, step A/B
R=/root/repo; O=$R/gpurun_out/r8w; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "text_encoder_as_one" 2>&1 | tail -15 > $O/tests.txt
timeout 600 python -m pytest tests/test_fullwidth_parity_gpu.py -x -q -k "rnn_encoder" 2>&1 | tail -3 >> $O/tests.txt
python tools/time_text.py 2>&1 | grep "text encoder" > $O/text.txt
for i in 1 2 3; do for v in on off; do
    echo -n "fused=$v  " >> $O/ab.txt
    if [ $v = off ]; then A="mogan_amd.attngan.model:RNN_ENCODER.FUSED=False"; else A="mogan_amd.attngan.model:RNN_ENCODER.FUSED=True"; fi
    MOGAN_CHAIN_EVENTS=1 timeout 600 python tools/ab_attr.py $A -- bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('host_enqueue_ms_per_step'), d.get('chain_ms'))" >> $O/ab.txt
done; done
