#!/bin/bash
cd /root/repo; O=gpurun_out/r6h; mkdir -p $O
for v in $VARIANTS; do
  echo "== $v"
  MOGAN_LIB=/root/repo/tools/lab/libmogan_$v.so timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-118 | head -3
done > $O/wino.txt 2>&1
