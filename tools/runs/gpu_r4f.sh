mkdir -p gpurun_out/r4f
python tools/time_pk.py > gpurun_out/r4f/pk_base.log 2>&1
MOGAN_LIB=$PWD/tools/lab/libmogan_pkA2.so python tools/time_pk.py > gpurun_out/r4f/pk_a2.log 2>&1
python -m pytest tests/test_kernels_gpu.py -q -x -k "packed or merged or deep" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -q -x -k "two_train_steps" 2>&1 | tail -3
tail -12 gpurun_out/r4f/pk_base.log; tail -12 gpurun_out/r4f/pk_a2.log
