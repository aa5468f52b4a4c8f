R=$GRAFT_REPO_ROOT; cd $R
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2; do
run A=1
run MOGAN_LAB_WRONG=1 MOGAN_LIB=$R/multiple-objects-gan_amd/build/lab_nobn.so
run MOGAN_LAB_WRONG=1 MOGAN_LIB=$R/multiple-objects-gan_amd/build/lab_nored.so
done
