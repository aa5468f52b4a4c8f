for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; done
python -m pytest tests/test_encoder_trainer_gpu.py tests/test_dp_engine_gpu.py -q -x 2>&1 | tail -3
python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-600
