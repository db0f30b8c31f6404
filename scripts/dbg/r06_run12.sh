B="--no-cpu-baseline --no-other-mode --no-other-configs --no-sustained --steps 60 --warmup 10"
for rep in 1 2; do for d in 0 1 2; do
ICS_FFT_DYNAMIC=$d python bench.py $B > gpurun_out/b12_4096_d${d}_$rep.json 2> gpurun_out/b12_4096_d${d}_$rep.err
done; done
