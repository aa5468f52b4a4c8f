R=$GRAFT_REPO_ROOT; cd $R
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2; do
run A=1
run MOGAN_WGRAD2=1
run MOGAN_WGRAD2=1 MOGAN_WGRAD2_BLOCKS=192
run MOGAN_WGRAD2=1 MOGAN_WGRAD2_BLOCKS=128
done
