#!/bin/bash
# Winograd forward lab: tile order (mb innermost, contiguous ranges) and the two role assignments of the SIMD's wave pair
cd /root/repo; O=gpurun_out/r6c; mkdir -p $O
for v in base ord r1 r2; do
  echo "== $v"
  MOGAN_LIB=/root/repo/tools/lab/libmogan_w6_$v.so timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids
done > $O/wino.txt 2>&1
