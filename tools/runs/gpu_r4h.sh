F='passed|failed|error|Error|assert'
python -m pytest tests/test_kernels_gpu.py -q -x -k "bilinear or logits_head" 2>&1 | grep -E "$F" | tail -5
python -m pytest tests/test_model_gpu.py tests/test_stackgan_gpu.py tests/test_encoder_trainer_gpu.py tests/test_fullwidth_parity_gpu.py -q -x 2>&1 | grep -E "$F" | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"
done
bash tools/prof_stats.sh r4h_ks_single > /dev/null 2>&1
