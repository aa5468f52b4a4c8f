python -m pytest tests/test_model_gpu.py -q -x -k "two_train_steps" > gpurun_out/gg_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/gg_tests.log | tail -5
for i in 1 2; do
MOGAN_G_GRAPHS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-250
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/gg_err.log | grep "^{" | cut -c1-250
done
for b in 4 8; do for g in 0 1; do echo "B=$b G graphs=$g"; MOGAN_G_GRAPHS=$g python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | cut -c1-250; done; done
for i in 1 2; do
MOGAN_G_GRAPHS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | cut -c1-200
MOGAN_G_WGRAD_FORK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>gpurun_out/gg_err.log | grep "^{" | cut -c1-200
MOGAN_CHAIN_EVENTS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('chain_ms'))"
MOGAN_G_GRAPHS=0 MOGAN_CHAIN_EVENTS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('chain_ms'))"
done
