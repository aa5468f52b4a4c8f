#!/bin/bash
# XCD-aware block / tile orders (wino5 super-tiles + non-temporal stores, wino_wgrad and dconv_wgrad blocks of one K range on one XCD)
R=/root/repo; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
for v in new old; do
  if [ $v = old ]; then export MOGAN_LIB=$R/tools/lab/libmogan_noxcd.so; else unset MOGAN_LIB; fi
  echo "== $v"; timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-40,119-170 | grep -v "^B[235] "
  timeout 300 python tools/time_dconv.py 2>/dev/null | grep "k4 s2\|k3 s1 up1" | cut -c1-40,85-130
done > $O/time.txt 2>&1
unset MOGAN_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or fp32_products or winograd" 2>&1 | tail -2 > $O/tests.txt
for i in 1 2; do for v in new old; do
    if [ $v = old ]; then export MOGAN_LIB=$R/tools/lab/libmogan_noxcd.so; else unset MOGAN_LIB; fi
    echo -n "$v " >> $O/ab.txt
    MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
done; done
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export MOGAN_LIB=$R/tools/lab/libmogan_noxcd.so; else unset MOGAN_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_$v -o w -- python $R/tools/pmc_wino.py > /dev/null 2>&1
  echo "$v FETCH_SIZE" >> $O/pmc.txt; python $R/tools/pmc_agg.py /tmp/p_$v | grep "wino\|wgrad" >> $O/pmc.txt
done
