# round 5: paired D pass -- tests + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "deep_block or bn_act" 2>&1 | grep -v "$F" | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | grep -v "$F" | tail -12
timeout 900 python -m pytest tests/test_fullwidth_parity_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | grep -v "$F" | tail -8
for i in 1 2; do
for p in 1 0; do
MOGAN_D_PAIR=$p python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pair=$p', round(d['value'],1), round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],1), 'parity', d.get('parity',{}).get('ok'))"
done; done
