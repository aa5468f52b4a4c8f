#!/bin/bash
# MFMA-busy per kernel of the single-stream step: one PMC pass (SQ + GRBM counters), aggregated per kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
E="MOGAN_FAST_INIT=1 MOGAN_STREAMS=0 MOGAN_WGRAD_STREAM=0 MOGAN_GRAPH_ENCODER=0"
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
env $E rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o m -- $B > /tmp/pm.log 2>&1
tail -3 /tmp/pm.log | cut -c1-200
python $R/tools/pmc_mfma.py /tmp/pm/m_counter_collection.csv /tmp/pm/m_kernel_trace.csv $O/${ROUND:-r06}_pmc_mfma.json
