mkdir -p gpurun_out/r5full
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > gpurun_out/r5full/pytest.log
tail -8 gpurun_out/r5full/pytest.log
python bench.py > gpurun_out/r5full/bench.log 2>gpurun_out/r5full/bench.err
tail -1 gpurun_out/r5full/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'roofline', r['kernel'], round(r['achieved'],1), round(r['frac'],3), 'parity ok', d['parity']['ok'], 'cpu', d['cpu_baseline']['value'])
for f in r['families']: print(f)
"
