mkdir -p gpurun_out/r4a
python -m pytest tests/test_model_gpu.py -q -x -k "two_train_steps" 2>&1 | tail -5 > gpurun_out/r4a/t_model.log
python -m pytest tests/test_fullsize_gpu.py -q -s -k "headline or secondary" 2>&1 | tail -25 > gpurun_out/r4a/t_fullsize.log
python -m pytest tests/test_fullwidth_parity_gpu.py -q -s -k "gnet_backward" 2>&1 | tail -15 > gpurun_out/r4a/t_gbwd.log
python -m pytest tests/test_encoder_trainer_gpu.py -q -k "bench_two_ranks" 2>&1 | tail -15 > gpurun_out/r4a/t_bench2.log
python bench.py > gpurun_out/r4a/bench0.log 2> gpurun_out/r4a/bench0.err
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r4a/ab_ready_$i.log 2>/dev/null
python bench.py --no-cpu-baseline --no-roofline --no-inputs-ready > gpurun_out/r4a/ab_noready_$i.log 2>/dev/null
done
python tools/time_wino.py > gpurun_out/r4a/time_wino.log 2>&1
tail -3 gpurun_out/r4a/t_*.log
