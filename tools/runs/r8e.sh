#!/bin/bash
# wino_wgrad2_kernel taken apart (512 blocks): lab builds without one kind of work each (wrong results; times only)
R=/root/repo; O=$R/gpurun_out/r8e; mkdir -p $O; cd $R
export MOGAN_WG2_BLOCKS=512
for v in product w2_NOSPLIT w2_NOMFMA w2_NOLD w2_NOST w2_NOCONS w2_NOSPLITMFMA w2_NOLDST product; do
  if [ $v = product ]; then unset MOGAN_LIB; else export MOGAN_LIB=$R/tools/lab/libmogan_$v.so; fi
  timeout 200 python tools/time_wgrad.py 2>&1 | grep "wgrad TF"
done > $O/time.txt 2>&1
