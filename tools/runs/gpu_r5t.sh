#!/bin/bash
# round 5: branches (D64, D128, D256, Inception) folded onto fewer of the engine's branch streams
cd /root/repo; mkdir -p gpurun_out/r5t
run() { echo -n "$* : "; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
( for i in 1 2 3; do run A=0; run MOGAN_BRANCH_MAP=0,1,2,0; done
run MOGAN_BRANCH_MAP=1,0,2,0
run MOGAN_BRANCH_MAP=0,1,2,0 MOGAN_FORCE_DIST=1
run A=0 MOGAN_FORCE_DIST=1 ) 2>&1 | tee gpurun_out/r5t/map2.txt
