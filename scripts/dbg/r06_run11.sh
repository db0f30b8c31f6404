cd image-cases-studies_amd/csrc
for cfg in "500 15 640" "4096 15" "2048 15" "4096 9"; do timeout 300 ./tools/bench_conv_fft $cfg > ../../gpurun_out/h11_$(echo $cfg | tr ' ' _).txt 2>&1; done
cd ../..
timeout 900 python -m pytest tests/test_gpu_fft.py -x -q -m gpu > gpurun_out/t11.txt 2>&1
B="--no-cpu-baseline --no-other-mode --no-other-configs --no-sustained --steps 60 --warmup 10"
python bench.py $B > gpurun_out/b11_4096_dyn.json 2> gpurun_out/b11_4096_dyn.err
ICS_FFT_DYNAMIC=0 python bench.py $B > gpurun_out/b11_4096_static.json 2> gpurun_out/b11_4096_static.err
python bench.py $B --mode nonblind --size 2048 > gpurun_out/b11_2048nb_dyn.json 2> gpurun_out/b11_2048nb_dyn.err
ICS_FFT_DYNAMIC=0 python bench.py $B --mode nonblind --size 2048 > gpurun_out/b11_2048nb_static.json 2> gpurun_out/b11_2048nb_static.err
python bench.py $B --size 2900 > gpurun_out/b11_2900_dyn.json 2> gpurun_out/b11_2900_dyn.err
ICS_FFT_DYNAMIC=0 python bench.py $B --size 2900 > gpurun_out/b11_2900_static.json 2> gpurun_out/b11_2900_static.err
