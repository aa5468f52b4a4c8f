F='passed|failed|error|Error|assert'
python -m pytest tests/test_stackgan_gpu.py -q -x 2>&1 | grep -E "$F" | tail -8
python -m pytest tests/test_fullsize_gpu.py -q -x -k "secondary" 2>&1 | grep -E "$F" | tail -5
for w in clevr mnist coco_s1 coco_s2; do python bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['value'],1), round(d['ms_per_step'],2))"; done
