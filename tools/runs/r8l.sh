#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8l; mkdir -p $O; cd $R
( time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > $O/gputests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gputests.txt 2>&1
