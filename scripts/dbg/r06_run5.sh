python scripts/ab_conv2.py > gpurun_out/ab_conv2_r06.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_perf_guard.py::test_auto_is_the_faster_choice_at_the_crossover_sizes -k "not (test_gpu_fft or test_gpu_rl or test_gpu_stages or test_banded or test_gpu_black or test_gpu_edges or test_gpu_bigpsf or test_gpu_baseline or test_gpu_deep)" > gpurun_out/t5_rest.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_perf_guard.py -x -q -m gpu > gpurun_out/t5_guard.txt 2>&1
