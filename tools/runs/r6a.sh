#!/bin/bash
# round 6, call a: the parked-weight-gradient regression test, a baseline bench line with the phase chain, per-layer Winograd times
cd /root/repo; O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "parked or two_train_steps" 2>&1 | tail -5 > $O/tests.txt
for i in 1 2; do MOGAN_CHAIN_EVENTS=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 >> $O/bench.jsonl; done
timeout 400 python tools/time_conv_layers.py $O/layers.csv > $O/layers.txt 2>&1
