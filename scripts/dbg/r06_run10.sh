timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_tv_mode.py -x -q -m gpu > gpurun_out/t10.txt 2>&1
B="--no-cpu-baseline --no-other-mode --no-other-configs --no-sustained --steps 40 --warmup 10"
for tv in 2 3; do
python bench.py $B --tv-mode $tv > gpurun_out/b10_4096_tv${tv}_auto.json 2> gpurun_out/b10_4096_tv${tv}_auto.err
python bench.py $B --tv-mode $tv --conv matrix > gpurun_out/b10_4096_tv${tv}_matrix.json 2> gpurun_out/b10_4096_tv${tv}_matrix.err
done
python bench.py $B --tv-mode 2 --size 2048 --mode nonblind > gpurun_out/b10_2048nb_tv2_auto.json 2> gpurun_out/b10_2048nb_tv2_auto.err
python bench.py $B --tv-mode 2 --size 2048 --mode nonblind --conv matrix > gpurun_out/b10_2048nb_tv2_matrix.json 2> gpurun_out/b10_2048nb_tv2_matrix.err
python bench.py $B --tv-mode 2 --size 2048 > gpurun_out/b10_2048_tv2_auto.json 2> gpurun_out/b10_2048_tv2_auto.err
python bench.py $B --tv-mode 2 --size 2048 --conv matrix > gpurun_out/b10_2048_tv2_matrix.json 2> gpurun_out/b10_2048_tv2_matrix.err
python bench.py $B --tv-mode 2 --size 2048 --conv fft > gpurun_out/b10_2048_tv2_fft.json 2> gpurun_out/b10_2048_tv2_fft.err
