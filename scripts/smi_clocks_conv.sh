T=image-cases-studies_amd/csrc/tools
for v in 0 a1; do   # `make -C image-cases-studies_amd/csrc tools` builds them
  echo "=== $v"
  ICS_BENCH_REPS=20000 $T/bench_conv_mfma_$v > gpurun_out/smi_$v.log 2>&1 &
  PID=$!
  sleep 2.0
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
  wait $PID
  cat gpurun_out/smi_$v.log | grep -v occup
done
