#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r8k; mkdir -p $O; cd $R
timeout 120 tools/lab/valu_rate > $O/valu_rate.txt 2>&1
