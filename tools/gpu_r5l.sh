R=$GRAFT_REPO_ROOT; cd $R
run() { pg=$1; shift
  if [ $pg = 1 ]; then EXTRA="MOGAN_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")"; else EXTRA="A=1"; fi
  env $EXTRA "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pg=$pg $*', round(d['value'],1), round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],1))"; }
for i in 1 2; do run 0 B=1; run 0 MOGAN_G_GRAPHS=0; run 1 B=1; run 1 MOGAN_G_GRAPHS=0; done
