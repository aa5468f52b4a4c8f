mkdir -p gpurun_out/r4q
python tools/aten_ops.py > gpurun_out/r4q/aten_ops.log 2>&1
grep -v "^/\|Warning\|warn" gpurun_out/r4q/aten_ops.log | head -70
