set -x
cd image-cases-studies_amd/csrc
for cfg in "500 15 640" "4096 15" "4096 9" "4096 21" "2048 15"; do timeout 300 ./tools/bench_conv_fft $cfg > ../../gpurun_out/h3_$(echo $cfg | tr ' ' _).txt 2>&1; done
timeout 300 ./tools/bench_conv_fft_m2o0 4096 15 > ../../gpurun_out/h3_o0_4096_15.txt 2>&1
cd ../..
timeout 900 python -m pytest tests/test_gpu_fft.py -x -q -m gpu > gpurun_out/t3_fft.txt 2>&1
B="--no-cpu-baseline --no-other-mode --no-other-configs --no-sustained --steps 60 --warmup 10"
python bench.py $B > gpurun_out/b3_4096_conv2.json 2> gpurun_out/b3_4096_conv2.err
ICS_FFT_CONV2=0 python bench.py $B > gpurun_out/b3_4096_two.json 2> gpurun_out/b3_4096_two.err
python bench.py $B --mode nonblind > gpurun_out/b3_4096nb_conv2.json 2> gpurun_out/b3_4096nb_conv2.err
ICS_FFT_CONV2=0 python bench.py $B --mode nonblind > gpurun_out/b3_4096nb_two.json 2> gpurun_out/b3_4096nb_two.err
