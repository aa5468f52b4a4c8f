#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8g; mkdir -p $O; cd $R
export MOGAN_WG2_BLOCKS=512
for v in product w2_NOSTY w2_NOSTX w2_NOLDSTY w2_NOLDSTX w2_NOLD w2_NOLDST; do
  if [ $v = product ]; then unset MOGAN_LIB; else export MOGAN_LIB=$R/tools/lab/libmogan_$v.so; fi
  timeout 200 python tools/time_wgrad.py 2>&1 | grep "wgrad TF"
done > $O/time.txt 2>&1
