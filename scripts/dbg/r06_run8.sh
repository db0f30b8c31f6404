set -x
cd image-cases-studies_amd/csrc
for cfg in "300 15 330" "4096 15" "6144 31" "2048 17"; do timeout 300 ./tools/bench_conv_fft $cfg > ../../gpurun_out/h8_$(echo $cfg | tr ' ' _).txt 2>&1; done
timeout 300 ./tools/bench_conv_fft_d2 4096 15 > ../../gpurun_out/h8_d2_4096_15.txt 2>&1; timeout 300 ./tools/bench_conv_fft_e0 4096 15 > ../../gpurun_out/h8_e0_4096_15.txt 2>&1
cd ../..
timeout 900 python -m pytest tests/test_gpu_fft.py -x -q -m gpu -k "update_inside or fused or one_unit" > gpurun_out/t8_fft.txt 2>&1
B="--no-cpu-baseline --no-other-mode --no-other-configs --no-sustained --steps 60 --warmup 10"
python bench.py $B > gpurun_out/b8_4096_upd.json 2> gpurun_out/b8_4096_upd.err
ICS_FFT_UPD=0 python bench.py $B > gpurun_out/b8_4096_noupd.json 2> gpurun_out/b8_4096_noupd.err
python bench.py $B --size 6144 --psf 31 > gpurun_out/b8_6144_upd.json 2> gpurun_out/b8_6144_upd.err
ICS_FFT_UPD=0 python bench.py $B --size 6144 --psf 31 > gpurun_out/b8_6144_noupd.json 2> gpurun_out/b8_6144_noupd.err
