# MOGAN_BRANCH_MAP: branches (D64, D128, D256, Inception) -> stream index; branches on one stream run in issue order (D256, Inception, D128, D64)
run() { echo "$*"; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ', round(d['value'],1))"; }
run A=0
run MOGAN_BRANCH_MAP=0,0,1,2
run MOGAN_BRANCH_MAP=1,1,0,1
run MOGAN_BRANCH_MAP=0,1,2,1
run MOGAN_BRANCH_MAP=0,1,2,0
run MOGAN_BRANCH_MAP=0,0,0,1
run MOGAN_BRANCH_MAP=2,1,0,3
run A=0
