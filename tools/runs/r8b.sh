#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8b; mkdir -p $O; cd $R
timeout 600 python tools/aten_ops.py > $O/aten.txt 2>&1
