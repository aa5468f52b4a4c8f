#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { v=$(env "$@" python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f' % d['value'])"); echo "$v  $*"; }
run A=0
run MOGAN_EARLY_DAMSM_BWD=0
run MOGAN_WINO22_MIN_TILES=256
run MOGAN_WINO22_MIN_TILES=512
run MOGAN_WINO22_MIN_TILES=768
run A=0
run MOGAN_EARLY_DAMSM_BWD=0 MOGAN_WINO22_MIN_TILES=512
run MOGAN_DSPLIT_FWD=256
run MOGAN_DSPLIT_FWD=768
run MOGAN_DSPLIT_WG=384
run MOGAN_DSPLIT_WG=1024
run A=0
run MOGAN_WINO22_DGRAD_MAXOW=32
run MOGAN_WINO22_DGRAD_MAXOW=8
run MOGAN_EARLY_DAMSM_BWD=0 MOGAN_WINO22_MIN_TILES=512
run A=0
