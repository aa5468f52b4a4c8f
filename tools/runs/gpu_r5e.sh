R=$GRAFT_REPO_ROOT; cd $R
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for i in 1 2; do timeout 900 python -m pytest tests/test_fullwidth_parity_gpu.py -q -x -s -k "dnet256_loss_backward" 2>&1 | grep -v "$F" | grep "rel-L2\|passed\|failed\|Error\|parity" | tail -5; done
git stash -q 2>/dev/null; echo stashed
