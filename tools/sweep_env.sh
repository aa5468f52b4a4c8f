#!/bin/bash
# A/B of dispatch knobs in ONE gpurun call (same box): img/s of the default bench per setting; the baseline is repeated
R=$GRAFT_REPO_ROOT; cd $R
run() { v=$(env "$@" python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f' % d['value'])"); echo "$v  $*"; }
run A=0
run MOGAN_WINO22=0
run MOGAN_WINO22_WGRAD=0 MOGAN_WINO22_DGRAD=0
run MOGAN_WINO22_WGRAD=0 MOGAN_WINO22_MIN_TILES=2048
run MOGAN_WINO22_WGRAD=0 MOGAN_WINO22_DGRAD=0 MOGAN_WINO22_MIN_TILES=2048
run MOGAN_WINO22_WGRAD=0 MOGAN_WINO22_MIN_TILES=4096
run A=0
run MOGAN_WINO22_WGRAD=0
run MOGAN_WINO22_WGRAD=0 MOGAN_DSPLIT_WG=1024
run MOGAN_WINO22_WGRAD=0 MOGAN_DSPLIT_WG=384
run MOGAN_WINO22=0 MOGAN_DSPLIT_FWD=768
run A=0
