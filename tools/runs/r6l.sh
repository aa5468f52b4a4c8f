#!/bin/bash
# in the step: dconv_wgrad_kernel with one / two blocks per CU asked of the allocator (alone on the GPU two is faster)
cd /root/repo; O=gpurun_out/r6l; mkdir -p $O
for i in 1 2; do
  for v in occ2 occ1; do
    if [ $v = occ1 ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_dwg1.so; else unset MOGAN_LIB; fi
    echo -n "$v " >> $O/ab.txt
    MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
  done
done
unset MOGAN_LIB
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd_prepared" 2>&1 | tail -2 > $O/tests.txt
