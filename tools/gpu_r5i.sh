R=$GRAFT_REPO_ROOT; cd $R
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],1))"; }
run A=1
run MOGAN_G_GRAPHS=1
run MOGAN_G_GRAPHS=1 MOGAN_G_WGRAD_FORK=1
run MOGAN_WINO_WG_BLOCKS=160
run MOGAN_WINO_WG_BLOCKS=224
run MOGAN_DSPLIT_WG=256
run MOGAN_SPLIT_TARGET=256
run MOGAN_SPLIT_TARGET=512
