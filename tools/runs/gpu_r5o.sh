#!/bin/bash
# round 5: dconv2 on 16 x 16 tiles and the space-to-depth forward of 4x4 s2 -- numerics, per-layer times, step A/B
cd /root/repo; mkdir -p gpurun_out/r5o; O=gpurun_out/r5o
MOGAN_WINO=0 timeout 600 python tools/check_dconv2.py > $O/check.txt 2>&1
tail -5 $O/check.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/time_dconv.py > $O/time_new.txt 2>&1
MOGAN_DCONV2=0 timeout 300 python tools/time_dconv.py > $O/time_old.txt 2>&1
paste -d'\n' $O/time_new.txt $O/time_old.txt | grep "k4 s2"
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"
  MOGAN_DCONV2=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s2d off', d['value'], d['ms_per_step'])"
done
