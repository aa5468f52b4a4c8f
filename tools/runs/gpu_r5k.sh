R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; RN=r05
cd /tmp && export TMPDIR=/tmp
E="MOGAN_FAST_INIT=1 MOGAN_STREAMS=0 MOGAN_WGRAD_STREAM=0 MOGAN_GRAPH_ENCODER=0"
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
env $E rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- $B > /tmp/pf.log 2>&1
env $E rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- $B > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $O/${RN}_pmc_traffic.json
