for cfg in "4 0" "4 2" "4 3" "4 0" "4 2" "4 1"; do set -- $cfg
echo "== single process, hw queues $1, reserved $2"
GPU_MAX_HW_QUEUES=$1 MOGAN_RESERVED_STREAMS=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric' | cut -c1-215
done
