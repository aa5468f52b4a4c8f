#!/bin/bash
# wino_wgrad2_kernel (blocks of 8 positions on raw LDS images, two per CU): values, per-layer time, block-count scan
R=/root/repo; O=$R/gpurun_out/r8d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or fp32_products or winograd or full_size" 2>&1 | tail -3 > $O/tests.txt
for v in "MOGAN_WG2=0" "MOGAN_WG2=1" "MOGAN_WG2_BLOCKS=256" "MOGAN_WG2_BLOCKS=320" "MOGAN_WG2_BLOCKS=448" "MOGAN_WG2_BLOCKS=512" "MOGAN_WG2_BLOCKS=768"; do
  echo -n "$v  "; env $v timeout 200 python tools/time_wgrad.py 2>&1 | grep "wgrad TF"
done > $O/time.txt 2>&1
