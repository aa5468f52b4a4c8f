#!/bin/bash
cd /root/repo; O=gpurun_out/r6v; mkdir -p $O
for c in -1 0 1 2; do echo "== cfg $c"; timeout 300 python tools/time_pk.py $c 2>&1 | grep -v amdgpu.ids | head -5 | cut -c1-120; done > $O/pk.txt 2>&1
