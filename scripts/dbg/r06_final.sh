timeout 1700 python -m pytest tests -q -m gpu -x > gpurun_out/final_tests.txt 2>&1
echo rc=$? >> gpurun_out/final_tests.txt
timeout 900 python -m pytest tests/test_gpu_deep_goldens.py tests/test_gpu_baseline_goldens.py -q -m gpu -s > gpurun_out/final_deep.txt 2>&1
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_driver_style.json 2> gpurun_out/final_bench_driver_style.err
ICS_COMMIT=a47bb32 bash scripts/collect_profiles_r06.sh 4096 15 blind > gpurun_out/prof_a.log 2>&1
ICS_COMMIT=a47bb32 bash scripts/collect_profiles_r06.sh 6144 31 blind > gpurun_out/prof_b.log 2>&1
python scripts/driver_timing.py 4096 15 20 > gpurun_out/final_driver_timing_4096.txt 2>&1
