#!/bin/bash
# wino5_fwd_kernel (16 waves per block) as the default 3x3 stride-1 forward / data gradient: kernel tests, then the step A/B
cd /root/repo; O=gpurun_out/r6i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or fp32_products or native or upsample" > $O/tests_full.txt 2>&1; tail -3 $O/tests_full.txt > $O/tests.txt
timeout 600 python -m pytest tests/test_fullwidth_parity_gpu.py -x -q -k "convolution_values or blocks" >> $O/tests_full.txt 2>&1; tail -3 $O/tests_full.txt >> $O/tests.txt
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_nowino5.so; else unset MOGAN_LIB; fi
    echo -n "$v " >> $O/ab.txt
    MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
  done
done
