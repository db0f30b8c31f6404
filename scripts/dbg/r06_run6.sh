python scripts/ab_fft.py 512,9 512,15 1024,9 1024,15 1024,17 1448,9 1448,13 1448,15 1448,17 2048,9 2048,11 2048,13 2048,15 2048,17 2900,9 2900,11 2900,13 2900,15 4096,5 4096,9 4096,11 4096,13 4096,15 6144,9 6144,15 > gpurun_out/ab_fft_r06b.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_deep_goldens.py tests/test_gpu_baseline_goldens.py -x -q -m gpu -s > gpurun_out/t6_deep.txt 2>&1
