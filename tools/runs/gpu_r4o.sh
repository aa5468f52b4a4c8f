F='passed|failed|error|Error|assert'
python -m pytest tests/test_encoder_trainer_gpu.py -q -x -k "cnn_encoder or graph" 2>&1 | grep -E "$F" | tail -5
for v in 0 1 0 1; do
MOGAN_INCEPTION_PANELS=$v python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('panels $v', d['value'], d['ms_per_step'])"
done
MOGAN_INCEPTION_PANELS=1 bash tools/prof_stats.sh r4o_ks_single > /dev/null 2>&1
