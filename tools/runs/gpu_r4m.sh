#!/bin/bash
# feasibility: Inception shapes on the packed-weight kernel
mkdir -p gpurun_out/r4m
for t in 384 768; do
  timeout 300 python tools/lab/pk_inception.py -1 0 $t > gpurun_out/r4m/pk_inc_t$t.log 2>&1
done
timeout 300 python tools/lab/pk_inception.py 0 1 384 > gpurun_out/r4m/pk_inc_c0_s1.log 2>&1
tail -20 gpurun_out/r4m/pk_inc_t384.log
