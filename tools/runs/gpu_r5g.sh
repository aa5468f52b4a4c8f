R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
tools/lab/mfma_peak > $O/r05_mfma_peak.txt 2>&1
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | grep -v "$F" | tail -4
timeout 900 python -m pytest tests/test_fullwidth_parity_gpu.py -q -x 2>&1 | grep -v "$F" | tail -4
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2; do
run A=1
run MOGAN_DCONV2=0
run MOGAN_WINO=0
run MOGAN_WINO_WGRAD=0
done
