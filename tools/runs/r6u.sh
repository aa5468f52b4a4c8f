#!/bin/bash
# pgemm_kernel<1,2>: two register sets of A operands at four waves per SIMD (lab) against three sets at three waves (product)
cd /root/repo; O=gpurun_out/r6u; mkdir -p $O
for v in occ3 occ4; do
  if [ $v = occ4 ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_pk4.so; else unset MOGAN_LIB; fi
  echo "== $v"; timeout 300 python tools/time_pk.py 2>&1 | grep -v amdgpu.ids
done > $O/pk.txt 2>&1
for i in 1 2; do for v in occ3 occ4; do
    if [ $v = occ4 ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_pk4.so; else unset MOGAN_LIB; fi
    echo -n "$v " >> $O/ab.txt
    MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
done; done
