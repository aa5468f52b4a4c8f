mkdir -p gpurun_out/r4e
bash tools/prof_stats.sh r4e_ks_single
mv gpurun_out/r4e_ks_single.csv gpurun_out/r4e/ks_single.csv
MOGAN_STREAMS=1 MOGAN_WGRAD_STREAM=1 MOGAN_GRAPH_ENCODER=1 bash tools/prof_stats.sh r4e_ks_multi
mv gpurun_out/r4e_ks_multi.csv gpurun_out/r4e/ks_multi.csv
MOGAN_CHAIN_EVENTS=1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r4e/chain.log
