# the bench as a member of a 1-rank RCCL group, repeated: every run must print its JSON line (stderr of a failing run is kept)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for i in 1 2 3 4 5 6 7 8; do
  Q=$(( (i % 3) + 2 )); RES=$(( (i % 2) * 3 ))
  env GPU_MAX_HW_QUEUES=$Q MOGAN_RESERVED_STREAMS=$RES MOGAN_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])") \
    timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/pg_$i.out 2> $O/pg_$i.err; rc=$?
  v=$(grep "^{" $O/pg_$i.out | python -c "import sys,json; print(round(json.loads(sys.stdin.readline())['value'],1))" 2>/dev/null)
  echo "run $i q=$Q res=$RES rc=$rc value=$v"
  if [ -z "$v" ]; then tail -15 $O/pg_$i.err; fi
done
