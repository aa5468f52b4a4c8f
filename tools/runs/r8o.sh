#!/bin/bash
# the N > 1 path of bench.py on the one-GPU box: two ranks share GPU 0 over gloo (everything but RCCL itself)
R=/root/repo; O=$R/gpurun_out/r8o; mkdir -p $O; cd $R
MOGAN_ONE_GPU=1 MOGAN_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 3 > $O/n2.log 2>&1
tail -1 $O/n2.log | cut -c1-1500 > $O/n2_line.txt
