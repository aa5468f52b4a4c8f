#!/bin/bash
# wino5 output transform in two passes of 48 channels (over V + X) against three passes of 32
cd /root/repo; O=gpurun_out/r6t; mkdir -p $O
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_ep3.so; else unset MOGAN_LIB; fi
  echo "== $v"; timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-118 | grep -v "^B[235] "
done; done > $O/wino.txt 2>&1
unset MOGAN_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or fp32_products or winograd" 2>&1 | tail -2 > $O/tests.txt
timeout 600 python -m pytest tests/test_fullwidth_parity_gpu.py -x -q -k "convolution_values or blocks" 2>&1 | tail -2 >> $O/tests.txt
