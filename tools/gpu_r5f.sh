R=$GRAFT_REPO_ROOT; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
MOGAN_WINO=0 python tools/check_dconv2.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Consider\|ef = " | tail -11
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or wino or native" 2>&1 | grep -v "$F" | tail -3
MOGAN_WINO=0 timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv2d_fwd_dgrad_wgrad or bf16_pipe" 2>&1 | grep -v "$F" | tail -3
