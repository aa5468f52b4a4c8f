# round 5: split D loss (real half early) -- tests + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | grep -v "$F" | tail -12
timeout 900 python -m pytest tests/test_fullwidth_parity_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | grep -v "$F" | tail -8
for i in 1 2; do
for p in 1 0; do
MOGAN_D_SPLIT=$p python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split=$p', round(d['value'],1), round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],1), 'parity', d.get('parity',{}).get('ok'))"
done; done
MOGAN_CHAIN_EVENTS=1 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('chain_ms'))"
