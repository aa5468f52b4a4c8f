#!/bin/bash
# lab: what bounds wino3_fwd_kernel -- builds with (a0) the filter loads of every K step reading step 0 (L1-hot), (notr) no input
# transform in the K loop, (nox) no halo loads / stores in the K loop; WRONG results, per-layer times only
cd /root/repo; mkdir -p gpurun_out/r5u
for v in "" noa nob nomma; do
  echo "== ${v:-product}"
  if [ -n "$v" ]; then export MOGAN_LIB=/root/repo/multiple-objects-gan_amd/build/lab_wino_$v.so; fi
  timeout 300 python tools/time_dconv.py 2>/dev/null | grep "k3 s1 up0" | head -3
done 2>&1 | tee gpurun_out/r5u/wino_lab2.txt
