R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B2="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
env MOGAN_FAST_INIT=1 MOGAN_STREAMS=0 MOGAN_WGRAD_STREAM=0 MOGAN_GRAPH_ENCODER=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $B2 > /tmp/ks.log 2>&1
cp /tmp/ks/ks_kernel_stats.csv $O/ks_kernel_stats.csv
python $R/tools/stats_summary.py $O/ks_kernel_stats.csv 13 80 > $O/per_step.txt 2>&1
head -3 $O/per_step.txt
