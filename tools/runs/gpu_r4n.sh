#!/bin/bash
mkdir -p gpurun_out/r4n
timeout 600 python tools/lab/panel_trunk_ab.py 4 one > gpurun_out/r4n/one.log 2>&1
tail -12 gpurun_out/r4n/one.log
