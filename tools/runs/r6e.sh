#!/bin/bash
# Winograd forward, fourth form: values + time per layer against the third form
cd /root/repo; O=gpurun_out/r6e; mkdir -p $O
for v in w4a; do
  echo "== $v"
  MOGAN_LIB=/root/repo/tools/lab/libmogan_$v.so timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids
  echo "== $v MOGAN_WINO4=0"
  MOGAN_WINO4=0 MOGAN_LIB=/root/repo/tools/lab/libmogan_$v.so timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | head -6
done > $O/wino.txt 2>&1
