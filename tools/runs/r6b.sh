#!/bin/bash
cd /root/repo; O=gpurun_out/r6b; mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -k "parked or two_train_steps" > $O/tests_full.txt 2>&1
grep -v "amdgpu.ids" $O/tests_full.txt | tail -40 > $O/tests.txt
