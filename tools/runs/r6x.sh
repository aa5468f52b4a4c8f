#!/bin/bash
# stream -> hardware-queue order and split target re-checked on the round-6 tree (kernel durations changed: wino5, dconv_wgrad)
cd /root/repo; O=gpurun_out/r6x; mkdir -p $O
run() { echo -n "$1 | " >> $O/ab.txt; env $2 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f img/s %.2f ms' % (d['value'], d['ms_per_step']))" >> $O/ab.txt; }
run "default (s2,s3,s1,wm,s0,gc,w2,w1,w0,cG,cD)" "A=1"
run "s2,s3,s1,wm,s0,gc,x,w2,w1,w0" "MOGAN_STREAM_ORDER=s2,s3,s1,wm,s0,gc,x,w2,w1,w0,cG,cD"
run "s2,s3,s1,wm,s0,gc,x,w2,x,w1,w0" "MOGAN_STREAM_ORDER=s2,s3,s1,wm,s0,gc,x,w2,x,w1,w0,cG,cD"
run "s2,s3,s1,wm,s0,gc,w2,w0,w1" "MOGAN_STREAM_ORDER=s2,s3,s1,wm,s0,gc,w2,w0,w1,cG,cD"
run "s2,s3,s1,wm,gc,s0,w2,w1,w0" "MOGAN_STREAM_ORDER=s2,s3,s1,wm,gc,s0,w2,w1,w0,cG,cD"
run "s2,s3,wm,s1,s0,gc,w2,w1,w0" "MOGAN_STREAM_ORDER=s2,s3,wm,s1,s0,gc,w2,w1,w0,cG,cD"
run "s2,s3,s1,gc,s0,wm,w2,w1,w0" "MOGAN_STREAM_ORDER=s2,s3,s1,gc,s0,wm,w2,w1,w0,cG,cD"
run "default again" "A=1"
run "MOGAN_SPLIT_TARGET=256" "MOGAN_SPLIT_TARGET=256"
run "MOGAN_SPLIT_TARGET=512" "MOGAN_SPLIT_TARGET=512"
run "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=3"
run "default again" "A=1"
