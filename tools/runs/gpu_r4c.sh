mkdir -p gpurun_out/r4c
rm -f gpurun_out/r4c/dbg.log
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv2d_fwd_dgrad_wgrad and force0" 2>&1 | tail -3 >> gpurun_out/r4c/dbg.log
for d in 0 2; do echo "== DBG $d" >> gpurun_out/r4c/dbg.log; MOGAN_WINO_DBG=$d python tools/time_wino.py 2>&1 | grep "96->192 128\|96-> 96 128\|96->192 64\|768->768" >> gpurun_out/r4c/dbg.log; done
cat gpurun_out/r4c/dbg.log
