#!/bin/bash
# Regenerates multiple-objects-gan_amd/hip/tuned_gemm_gfx950.csv on an MI355X: per workload one eager single-stream step is
# profiled per launch (bench.py, heuristic dispatch), every GEMM shape of it is timed under all (tile config, split) pairs
# (tools/tune_gemm.py), and the winners that beat the heuristic by >= 3 % are merged (tools/make_tuned_table.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export MOGAN_TUNED=0
for b in 16 8 4 32; do
  MOGAN_LAYERS_CSV=$PWD/gpurun_out/layers_attngan_b$b.csv python bench.py --batch $b --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python tools/tune_gemm.py gpurun_out/layers_attngan_b$b.csv gpurun_out/tune_attngan_b$b.csv | head -1
done
for w in mnist clevr coco_s1 coco_s2; do
  MOGAN_LAYERS_CSV=$PWD/gpurun_out/layers_$w.csv python bench.py --workload $w --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python tools/tune_gemm.py gpurun_out/layers_$w.csv gpurun_out/tune_$w.csv | head -1
done
python tools/make_tuned_table.py gpurun_out/tune_*.csv
