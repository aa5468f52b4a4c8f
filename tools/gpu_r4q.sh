python -m pytest tests/test_kernels_gpu.py -q -x -k "panel" 2>&1 | tail -15
