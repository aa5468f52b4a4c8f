#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5w
for v in 1 2 3 0; do echo "-- MOGAN_STEM=$v"; MOGAN_STEM=$v timeout 300 python tools/check_stem.py 2>&1 | grep -v amdgpu.ids | grep "B16\|B24\|worst"; done | tee gpurun_out/r5w/stem.txt
