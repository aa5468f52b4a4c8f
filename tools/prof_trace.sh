#!/bin/bash
# kernel trace (timestamps) of the default multi-stream bench -> gpurun_out/$1_kernel_trace.csv + timeline summary
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
NAME=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
env MOGAN_FAST_INIT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/$NAME -o tr -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline "$@" > /tmp/$NAME.log 2>&1
f=$(ls /tmp/$NAME/*kernel_trace.csv | head -1)
cp $f $O/${NAME}_kernel_trace.csv 2>/dev/null; python $R/tools/timeline.py $f | tee $O/${NAME}_timeline.txt; python $R/tools/timeline2.py $f 7 | tee -a $O/${NAME}_timeline.txt; python $R/tools/timeline2.py $f 8 | tee -a $O/${NAME}_timeline.txt
