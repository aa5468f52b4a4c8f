#!/bin/bash
# Regenerates the judged artefacts under gpurun_out/ on the GPU box (copy them to profiles/ afterwards):
#   <round>_pmc_traffic.json, ${RN}_bench_kernel_stats{,_multistream}.csv, ${RN}_bench_n1.json.log, ${RN}_family_bench_n1.jsonl
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
RN=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp
E="MOGAN_FAST_INIT=1 MOGAN_STREAMS=0 MOGAN_WGRAD_STREAM=0 MOGAN_GRAPH_ENCODER=0"
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
env $E rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- $B > /tmp/pf.log 2>&1
env $E rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- $B > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $R/profiles/${RN}_pmc_traffic.json > /tmp/pt.log 2>&1
cp $R/profiles/${RN}_pmc_traffic.json $O/
B2="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
# MOGAN_FAST_INIT=1: plain normal_ initialisation instead of orthogonal_ (rocSOLVER QR + Tensile GEMMs on the GPU were 82 % of the
# round-4 summary); tools/stats_summary.py turns the csv into the per-step table <round>_bench_kernel_stats_per_step.txt
env MOGAN_FAST_INIT=1 MOGAN_STREAMS=0 MOGAN_WGRAD_STREAM=0 MOGAN_GRAPH_ENCODER=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $B2 > /tmp/ks.log 2>&1
cp /tmp/ks/ks_kernel_stats.csv $O/${RN}_bench_kernel_stats.csv
python $R/tools/stats_summary.py $O/${RN}_bench_kernel_stats.csv 13 > $O/${RN}_bench_kernel_stats_per_step.txt 2>&1
env MOGAN_FAST_INIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/km -o km -- $B2 > /tmp/km.log 2>&1
cp /tmp/km/km_kernel_stats.csv $O/${RN}_bench_kernel_stats_multistream.csv
cd $R
python bench.py --steps 20 --warmup 5 > $O/${RN}_bench_n1.json.log 2>/dev/null
: > $O/${RN}_family_bench_n1.jsonl
for w in mnist clevr coco_s1 coco_s2; do python bench.py --workload $w --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/${RN}_family_bench_n1.jsonl; done
tail -1 $O/${RN}_bench_n1.json.log | cut -c1-200; wc -l $O/${RN}_family_bench_n1.jsonl; ls -la $O/${RN}_*
