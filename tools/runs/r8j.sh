#!/bin/bash
# wino_wgrad2_kernel, second version (half-transformed pair-ready dY image): values, per-layer time, block-count scan
R=/root/repo; O=$R/gpurun_out/r8j; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or fp32_products or winograd or full_size" 2>&1 | tail -3 > $O/tests.txt
for v in "MOGAN_WG2=0" "MOGAN_WG2_BLOCKS=384" "MOGAN_WG2_BLOCKS=448" "MOGAN_WG2_BLOCKS=512"; do
  echo -n "$v  "; env $v timeout 200 python tools/time_wgrad.py 2>&1 | grep "wgrad TF"
done > $O/time.txt 2>&1
