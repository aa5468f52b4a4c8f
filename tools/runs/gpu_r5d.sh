R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullwidth_parity_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -q 2>&1 | grep -v "$F" | tail -12
MOGAN_D_SPLIT=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -k "two_train_steps" 2>&1 | grep -v "$F" | tail -4
MOGAN_D_PAIR=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -k "two_train_steps or losses" 2>&1 | grep -v "$F" | tail -4
MOGAN_D_PAIR=1 MOGAN_CHAIN_EVENTS=1 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pair', d.get('value'), d.get('chain_ms'))"
