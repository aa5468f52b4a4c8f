#!/bin/bash
# dconv_wgrad_kernel: blocks per CU asked of the register allocator (1 = round 5, 2 = product, 3)
cd /root/repo; O=gpurun_out/r6j; mkdir -p $O
for v in 1 2 3; do
  if [ $v = 2 ]; then unset MOGAN_LIB; else export MOGAN_LIB=/root/repo/tools/lab/libmogan_dwg$v.so; fi
  echo "== occupancy hint $v"
  timeout 300 python tools/time_dconv.py 2>/dev/null | grep "k4 s2\|k3 s1 up1\|3072->" | cut -c1-40,85-130
done > $O/wg.txt 2>&1
unset MOGAN_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d" > $O/tests_full.txt 2>&1; tail -2 $O/tests_full.txt > $O/tests.txt
