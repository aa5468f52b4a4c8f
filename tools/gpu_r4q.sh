python -m pytest tests/test_kernels_gpu.py -q -x -k "bn_act or deep_block or group_sum" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_parity_gpu.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
bash tools/prof_stats.sh r4s_ks_single > /dev/null 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; done
