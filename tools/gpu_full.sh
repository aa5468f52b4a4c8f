mkdir -p gpurun_out/r4full
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > gpurun_out/r4full/pytest.log
tail -6 gpurun_out/r4full/pytest.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dm -o ks -- python $GRAFT_REPO_ROOT/tools/time_damsm.py > /tmp/dm.log 2>&1; cp /tmp/dm/ks_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r4full/damsm_stats.csv
