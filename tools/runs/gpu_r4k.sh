F='passed|failed|error|Error|assert'
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "damsm or words or sent or losses" 2>&1 | grep -E "$F" | tail -5
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dm -o ks -- python $GRAFT_REPO_ROOT/tools/time_damsm.py > /tmp/dm.log 2>&1; grep "damsm_words" /tmp/dm/ks_kernel_stats.csv | cut -c1-60,200-330
tail -1 /tmp/dm.log
