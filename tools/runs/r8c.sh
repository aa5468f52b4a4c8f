#!/bin/bash
# wino_wgrad_kernel taken apart: lab builds without one kind of work each (wrong results; times only)
R=/root/repo; O=$R/gpurun_out/r8c; mkdir -p $O; cd $R
for v in product wg_NOSPLIT wg_NOMFMA wg_NOQ wg_NOV wg_NOLD wg_NOQV wg_NOSPLITMFMA wg_ONLYMFMA product; do
  if [ $v = product ]; then unset MOGAN_LIB; else export MOGAN_LIB=$R/tools/lab/libmogan_$v.so; fi
  timeout 200 python tools/time_wgrad.py 2>&1 | grep "wgrad TF"
done > $O/time.txt 2>&1
