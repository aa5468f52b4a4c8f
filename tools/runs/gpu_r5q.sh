#!/bin/bash
# lab: upper bound of taking the BatchNorm forward statistics from the convolution epilogues -- a build whose bn_partial_kernel
# reads 1/16 of its range (sane but WRONG statistics) against the product build, same box
cd /root/repo; mkdir -p gpurun_out/r5q; O=gpurun_out/r5q
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2; do
  run "default"
  MOGAN_LIB=/root/repo/multiple-objects-gan_amd/build/lab_bnsub.so run "bn_partial 1/16"
  MOGAN_STREAMS=0 run "single-stream default"
  MOGAN_STREAMS=0 MOGAN_LIB=/root/repo/multiple-objects-gan_amd/build/lab_bnsub.so run "single-stream bn_partial 1/16"
done 2>&1 | tee $O/ab.txt
