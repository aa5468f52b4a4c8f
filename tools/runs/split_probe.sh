# block-count target of the direct weight-gradient kernels' pixel-tile split, in the step (img/s); default 640
run() { echo "$*"; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ', round(d['value'],1))"; }
for r in 1 2; do run A=0; run MOGAN_DSPLIT_WG=256; run MOGAN_DSPLIT_WG=320; run MOGAN_DSPLIT_WG=384; run MOGAN_DSPLIT_WG=448; done
