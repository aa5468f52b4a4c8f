#!/bin/bash
# wino5: XCD-contiguous tile ranges and non-temporal output stores -- time per layer and L2-miss bytes (FETCH_SIZE / WRITE_SIZE)
R=/root/repo; O=$R/gpurun_out/r6y; mkdir -p $O
cd $R
for v in w5b w5x w5xn; do
  echo "== $v"; MOGAN_LIB=$R/tools/lab/libmogan_$v.so timeout 300 python tools/check_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-100 | grep -v "^B[235] "
done > $O/time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in w5b w5x w5xn; do for c in FETCH_SIZE WRITE_SIZE; do
  MOGAN_LIB=$R/tools/lab/libmogan_$v.so timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_${v}_$c -o w -- python $R/tools/pmc_wino.py > /dev/null 2>&1
  echo "$v $c" >> $O/pmc.txt; python $R/tools/pmc_agg.py /tmp/p_${v}_$c | grep wino5 >> $O/pmc.txt
done; done
