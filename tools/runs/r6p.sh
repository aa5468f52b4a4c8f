#!/bin/bash
# wino_wgrad16_kernel (16 waves, one position each) against wino_wgrad_kernel: values + time per layer, kernel tests, step A/B
cd /root/repo; O=gpurun_out/r6p; mkdir -p $O
for v in new old; do
  if [ $v = old ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_nowg16.so; else unset MOGAN_LIB; fi
  echo "== $v"; timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-40,119-170
done > $O/wino.txt 2>&1
unset MOGAN_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or fp32_products" > $O/t1.txt 2>&1; tail -3 $O/t1.txt > $O/tests.txt
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_nowg16.so; else unset MOGAN_LIB; fi
    echo -n "$v " >> $O/ab.txt
    MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
  done
done
