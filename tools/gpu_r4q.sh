mkdir -p gpurun_out/r4q
P=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
MOGAN_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$P python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r4q/pg.out 2> gpurun_out/r4q/pg.err
echo rc=$?
tail -5 gpurun_out/r4q/pg.err | cut -c1-300
tail -1 gpurun_out/r4q/pg.out | cut -c1-100
