#!/bin/bash
# Everything under profiles/<round>_* that is measured (run on the GPU box; copy gpurun_out/<round>_* to profiles/ afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; RN=${ROUND:-r06}; export ROUND=$RN; cd $R
ROUND=$RN bash tools/refresh_profiles.sh
MOGAN_LAYERS_CSV=$O/${RN}_layers_single_stream.csv python bench.py --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_mfma.sh
bash tools/prof_trace.sh ${RN}_trace > /dev/null 2>&1
python tools/main_chain.py $O/${RN}_trace_kernel_trace.csv 7 >> $O/${RN}_trace_timeline.txt 2>&1; rm -f $O/${RN}_trace_kernel_trace.csv
: > $O/${RN}_batch_sizes.jsonl
for b in 4 8 16 32; do python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $O/${RN}_batch_sizes.jsonl; done
ls -la $O/${RN}_*
