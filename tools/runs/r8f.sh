#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8f; mkdir -p $O; cd $R
export MOGAN_WG2_BLOCKS=512
for d in randn zeros ones small; do for v in 1 0; do
  echo -n "data=$d WG2=$v  "; TW_DATA=$d MOGAN_WG2=$v timeout 200 python tools/time_wgrad.py 2>&1 | grep "wgrad TF"
done; done > $O/time.txt 2>&1
