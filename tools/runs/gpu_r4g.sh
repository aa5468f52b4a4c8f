mkdir -p gpurun_out/r4g
F='passed|failed|error|Error|assert'
python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd_prepared or conv2d_fwd_dgrad_wgrad or cat_channels or stn_shared" 2>&1 | grep -E "$F" | tail -5
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_parity_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | grep -E "$F" | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prep', d['value'], d['ms_per_step'])"
MOGAN_WINO_PREP=0 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noprep', d['value'], d['ms_per_step'])"
MOGAN_WINO_PREP=0 MOGAN_WINO_V=2 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v2', d['value'], d['ms_per_step'])"
done
