#!/bin/bash
# round 6, third session: sanity of the rebuilt tree + the packed-weight GEMM per layer (tile / split variants) + its FETCH_SIZE per layer
R=/root/repo; O=$R/gpurun_out/r8a; mkdir -p $O; cd $R
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-400 > $O/bench.txt
for a in "-1 0" "1 0" "-1 4" "-1 8" "1 4"; do echo "== cfg split: $a"; timeout 300 python tools/time_pk.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-36,66-82,100-116; done > $O/time_pk.txt 2>&1
bash tools/runs/r7a.sh; cp $R/gpurun_out/r7a/pk.txt $O/pk_fetch.txt
