mkdir -p gpurun_out/r4p
MOGAN_LAYERS_CSV=$PWD/gpurun_out/r4p/layers.csv python bench.py --no-cpu-baseline > gpurun_out/r4p/bench.log 2>&1
tail -1 gpurun_out/r4p/bench.log | cut -c1-400
