#!/bin/bash
# final tree: smoke(), the whole -m gpu suite, then the profile refresh
cd /root/repo; O=gpurun_out/r6w; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 3000 python -m pytest tests -x -q -m gpu > $O/full.txt 2>&1; grep "passed\|failed" $O/full.txt | tail -2 > $O/tests.txt
ROUND=r06 bash tools/refresh_all.sh > gpurun_out/r06_refresh.log 2>&1
