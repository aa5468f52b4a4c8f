#!/bin/bash
# default bench run with the per-phase wall clock (where did the driver's 491 s go?), before / after the CholeskyQR2 orthogonal init
cd /root/repo; O=gpurun_out/r6r; mkdir -p $O
timeout 300 python -m pytest tests/test_encoder_trainer_gpu.py -x -q -k orthogonal 2>&1 | tail -2 > $O/tests.txt
( time python bench.py > $O/bench.json ) 2> $O/bench.err
grep "^\[bench\]\|^real" $O/bench.err > $O/phases.txt
