for i in 1 2; do
MOGAN_G_GRAPHS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | cut -c1-200
MOGAN_G_WGRAD_FORK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>gpurun_out/gg_err.log | grep "^{" | cut -c1-200
MOGAN_CHAIN_EVENTS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('chain_ms'))"
MOGAN_G_GRAPHS=0 MOGAN_CHAIN_EVENTS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('chain_ms'))"
done
