#!/bin/bash
# rocprofv3 kernel stats of the bench command (single-stream mode by default): $1 = output name under gpurun_out/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
NAME=${1:-ks}; shift
cd /tmp && export TMPDIR=/tmp
env MOGAN_FAST_INIT=1 MOGAN_STREAMS=${MOGAN_STREAMS:-0} MOGAN_WGRAD_STREAM=${MOGAN_WGRAD_STREAM:-0} MOGAN_GRAPH_ENCODER=${MOGAN_GRAPH_ENCODER:-0} \
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$NAME -o ks -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > /tmp/$NAME.log 2>&1
cp /tmp/$NAME/ks_kernel_stats.csv $O/$NAME.csv
tail -2 /tmp/$NAME.log | cut -c1-300
