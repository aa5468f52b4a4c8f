#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8r; mkdir -p $O; cd $R
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -x -q -k "dconv2_prepared" 2>&1 | grep -v "Extension modules" | head -60 > $O/t.txt
