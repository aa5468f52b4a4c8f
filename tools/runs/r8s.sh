#!/bin/bash
# the up-convolutions' K = T w T^t (and dconv2 image) kept by the optimizer: tests, step A/B
R=/root/repo; O=$R/gpurun_out/r8s; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "upconv or upsample or prepared" 2>&1 | grep "passed\|failed\|Error\|assert" | head -8 > $O/tests.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | head -8 >> $O/tests.txt
for i in 1 2 3; do for v in on off; do
    echo -n "owner-kept=$v  " >> $O/ab.txt
    if [ $v = off ]; then A="mogan_amd.hip.ops:UPCONV_OWNED=False"; else A="mogan_amd.hip.ops:UPCONV_OWNED=True"; fi
    MOGAN_CHAIN_EVENTS=1 timeout 600 python tools/ab_attr.py $A -- bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
done; done
