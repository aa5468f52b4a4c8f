#!/bin/bash
# prepared dconv2 filter images owned by the optimizer (mogan_conv_prep_bytes / _group): tests, launches per step, step A/B
R=/root/repo; O=$R/gpurun_out/r8q; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "prepared or conv2d_fwd_dgrad_wgrad or packed_weight" 2>&1 | tail -4 > $O/tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "two_train_steps or resume or parked" 2>&1 | tail -4 >> $O/tests.txt
for i in 1 2 3; do for v in on off; do
    echo -n "prep=$v  " >> $O/ab.txt
    if [ $v = off ]; then A="mogan_amd.hip.ops:D2_PREP=False"; else A="mogan_amd.hip.ops:D2_PREP=True"; fi
    MOGAN_CHAIN_EVENTS=1 timeout 600 python tools/ab_attr.py $A -- bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
done; done
