#!/bin/bash
# prepared Winograd filter images owned by FlatAdam: kernel tests, engine tests, step A/B against the per-call transform
cd /root/repo; O=gpurun_out/r6k; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd or conv2d or packed" > $O/t1.txt 2>&1; tail -4 $O/t1.txt > $O/tests.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q > $O/t2.txt 2>&1; tail -4 $O/t2.txt >> $O/tests.txt
for i in 1 2; do
  for v in prep percall; do
    echo -n "$v " >> $O/ab.txt
    if [ $v = percall ]; then A="mogan_amd.hip.ops:WINO_PREP=False"; else A=""; fi
    MOGAN_CHAIN_EVENTS=1 timeout 600 python tools/ab_attr.py $A -- bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']), d.get('chain_ms'))" >> $O/ab.txt
  done
done
