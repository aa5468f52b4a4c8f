#!/bin/bash
# the whole -m gpu suite on the round-6 tree
cd /root/repo; O=gpurun_out/r6o; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/full.txt 2>&1; tail -5 $O/full.txt > $O/tests.txt
