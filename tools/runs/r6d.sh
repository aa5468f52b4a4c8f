#!/bin/bash
# Winograd forward lab: one wave of every SIMD's pair starts each K step late (s_sleep) -- 4/8/12/16 x 64 cycles, two role maps
cd /root/repo; O=gpurun_out/r6d; mkdir -p $O
for v in ord s1_4 s1_8 s1_12 s1_16 s2_8; do
  echo "== $v"
  MOGAN_LIB=/root/repo/tools/lab/libmogan_w6_$v.so timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | head -6
done > $O/wino.txt 2>&1
