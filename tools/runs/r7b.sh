#!/bin/bash
cd /root/repo; O=gpurun_out/r7b; mkdir -p $O
for i in 1 2; do for v in base prio; do
  if [ $v = prio ]; then export MOGAN_LIB=/root/repo/tools/lab/libmogan_w5prio.so; else unset MOGAN_LIB; fi
  echo "== $v"; timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-100 | grep -v "^B[235] "
done; done > $O/time.txt 2>&1
