#!/bin/bash
# round 5: does a faster kernel show in the step?  single-stream and multi-stream A/B of the dconv2 levels
cd /root/repo; mkdir -p gpurun_out/r5p; O=gpurun_out/r5p
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2; do
  for lv in 2 1 0; do
    MOGAN_STREAMS=0 MOGAN_DCONV2=$lv run "single-stream dconv2=$lv"
  done
  for lv in 2 1 0; do
    MOGAN_DCONV2=$lv run "multi-stream dconv2=$lv"
  done
done 2>&1 | tee $O/ab.txt
