mkdir -p gpurun_out/r4b
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv2d_fwd_dgrad_wgrad or cat_channels or stn or bf16_pipe" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > gpurun_out/r4b/t_kernels.log
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_parity_gpu.py -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > gpurun_out/r4b/t_model.log
python tools/time_wino.py > gpurun_out/r4b/time_wino.log 2>&1
MOGAN_WINO_V=2 python tools/time_wino.py > gpurun_out/r4b/time_wino_v2.log 2>&1
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r4b/bench_a.log 2>/dev/null
MOGAN_WINO_V=2 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r4b/bench_v2.log 2>/dev/null
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r4b/bench_b.log 2>/dev/null
python tools/aten_ops.py > gpurun_out/r4b/aten_ops.log 2>&1
tail -n 4 gpurun_out/r4b/t_*.log gpurun_out/r4b/time_wino.log
