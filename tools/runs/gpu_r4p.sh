mkdir -p gpurun_out/r4p
for t in 512 256 128 1 768 512; do
MOGAN_PT_TARGET=$t python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('target $t', d['value'], d['ms_per_step'])"
done
for t in 128 1; do
MOGAN_PT_TARGET=$t MOGAN_LAYERS_CSV=$PWD/gpurun_out/r4p/layers_pt$t.csv python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-10
done
