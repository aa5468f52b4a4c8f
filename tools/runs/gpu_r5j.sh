R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
run() { pg=$1; ord=$2; res=$3
  if [ $pg = 1 ]; then EXTRA="MOGAN_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")"; else EXTRA="A=1"; fi
  v=$(env MOGAN_STREAM_ORDER=$ord GPU_MAX_HW_QUEUES=4 MOGAN_RESERVED_STREAMS=$res $EXTRA python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f,%.2f' % (d['value'], d['ms_per_step']))")
  echo "pg=$pg,res=$res,$ord,$v" | tee -a $O/order3.csv; }
: > $O/order3.csv
for rep in 1 2; do for ord in "s2,s3,s1,wm,s0" "s2,s3,wm,s1,s0" "s2,s3,s1,wm,s0,w2,w1,w0,cG,cD"; do run 0 $ord 0; run 1 $ord 0; done; done
run 1 "s2,s3,s1,wm,s0" 3
