R=$GRAFT_REPO_ROOT; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv2d_fwd_dgrad_wgrad or lrelu" 2>&1 | grep -v "$F" | tail -3
python tools/time_smallc.py 2>&1 | grep -v amdgpu.ids | tail -2
MOGAN_SC_K4S2=0 python tools/time_smallc.py 2>&1 | grep -v amdgpu.ids | tail -2
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
run A=1; run MOGAN_SC_K4S2=0; run A=1; run MOGAN_SC_K4S2=0
