set -x
cd image-cases-studies_amd/csrc
for cfg in "4096 15" "6144 31" "4096 45" "2048 21"; do timeout 300 ./tools/bench_conv_fft $cfg > ../../gpurun_out/h_$(echo $cfg | tr ' ' _).txt 2>&1; done
cd ../..
timeout 900 python -m pytest tests/test_gpu_fft.py -x -q -m gpu > gpurun_out/t_fft.txt 2>&1
B="--no-cpu-baseline --no-other-mode --no-other-configs --no-sustained --steps 60 --warmup 10"
python bench.py $B --conv fft > gpurun_out/b4096_fft_fused.json 2> gpurun_out/b4096_fft_fused.err
ICS_FFT_FUSED=0 python bench.py $B --conv fft > gpurun_out/b4096_fft_two.json 2> gpurun_out/b4096_fft_two.err
python bench.py $B > gpurun_out/b4096_auto.json 2> gpurun_out/b4096_auto.err
python bench.py $B --size 6144 --psf 31 > gpurun_out/b6144_fused.json 2> gpurun_out/b6144_fused.err
ICS_FFT_FUSED=0 python bench.py $B --size 6144 --psf 31 > gpurun_out/b6144_two.json 2> gpurun_out/b6144_two.err
python bench.py $B --size 4096 --psf 45 > gpurun_out/b4096_45.json 2> gpurun_out/b4096_45.err
