timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/fuzz.txt 2>&1; tail -5 gpurun_out/fuzz.txt
