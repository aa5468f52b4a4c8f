# round 5, first call: baseline of the inherited tree + the 2B estimate
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-600
python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-300
MOGAN_CHAIN_EVENTS=1 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 > $O/chain.log; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r5a/chain.log').read()); print(d.get('value'), d.get('chain_ms'))
PY
python tools/time_dpair.py 2>&1 | tail -3
