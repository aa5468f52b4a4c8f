F='passed|failed|error|Error|assert'
python -m pytest tests/test_kernels_gpu.py -q -x -k "damsm or words or sent" 2>&1 | grep -E "$F" | tail -5
python -m pytest tests/test_model_gpu.py -q -x -k "losses or two_train_steps" 2>&1 | grep -E "$F" | tail -5
python tools/time_damsm.py 2>&1 | tail -8
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"
done
