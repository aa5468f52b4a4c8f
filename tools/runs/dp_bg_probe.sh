# N>1 code path on the 1-GPU box: the two-rank gloo tests, then bench.py as a member of a 1-rank RCCL group for the
# (hardware queues, reserved streams) choices, discriminator branches as hipGraphs (MOGAN_BRANCH_GRAPHS_DP=1) or eager
mkdir -p gpurun_out
python -m pytest tests/test_dp_engine_gpu.py tests/test_model_gpu.py -q -x -k "two_rank or rccl or two_train_steps" > gpurun_out/dp_tests.log 2>&1; tail -3 gpurun_out/dp_tests.log
for cfg in "1 4 3" "1 4 0" "1 4 2" "1 3 3" "0 4 3" "0 3 3" "1 4 3"; do set -- $cfg
echo "== RCCL world-1, branch graphs DP=$1, hw queues $2, reserved $3"
MOGAN_BRANCH_GRAPHS_DP=$1 GPU_MAX_HW_QUEUES=$2 MOGAN_RESERVED_STREAMS=$3 MOGAN_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29411 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric' | cut -c1-215
done
