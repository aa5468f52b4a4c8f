#!/bin/bash
# img/s of the default bench step under GPU_MAX_HW_QUEUES x reserved streams x (with / without the RCCL process group at world = 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
echo "queues,reserved,process_group,img_per_s,ms_per_step" > $O/${ROUND:-r04}_queue_table.csv
for q in 2 3 4; do for res in 0 2 3; do for pg in 0 1; do
  if [ $pg = 1 ]; then EXTRA="MOGAN_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")"; else EXTRA=""; fi
  v=$(env GPU_MAX_HW_QUEUES=$q MOGAN_RESERVED_STREAMS=$res $EXTRA python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f,%.2f' % (d['value'], d['ms_per_step']))")
  [ -z "$v" ] && v="failed,"        # (a process-group cell whose random rendezvous port was still in use; the port is now asked from the OS)
  echo "$q,$res,$pg,$v" | tee -a $O/${ROUND:-r04}_queue_table.csv
done; done; done
