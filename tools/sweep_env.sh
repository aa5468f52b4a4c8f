#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { v=$(env "$@" python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f' % d['value'])"); echo "$v  $*"; }
run A=0
run MOGAN_BRANCH_MAP=0,0,1,0
run MOGAN_BRANCH_MAP=0,0,1,2
run MOGAN_BRANCH_MAP=0,1,2,0
run MOGAN_BRANCH_MAP=0,0,0,1
run A=0
run MOGAN_BRANCH_MAP=0,0,1,0 MOGAN_RESERVED_STREAMS=0
run MOGAN_BRANCH_MAP=0,0,1,0 GPU_MAX_HW_QUEUES=4 MOGAN_RESERVED_STREAMS=0
run MOGAN_BRANCH_MAP=0,0,1,2 GPU_MAX_HW_QUEUES=4 MOGAN_RESERVED_STREAMS=0
run MOGAN_BRANCH_MAP=0,0,0,0
run A=0
