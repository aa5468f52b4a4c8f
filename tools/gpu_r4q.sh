python -m pytest tests/test_model_gpu.py tests/test_fullwidth_parity_gpu.py tests/test_stackgan_gpu.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; done
