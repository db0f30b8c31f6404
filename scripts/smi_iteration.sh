# board power and clocks while bench.py's iteration runs (rocm-smi samples during a long timed region)
for mode in blind nonblind; do
  echo "=== $mode"
  python3 bench.py --mode $mode --steps 6000 --warmup 50 --no-cpu-baseline --no-other-configs --no-other-mode --no-profile > gpurun_out/smi_$mode.json 2>/dev/null &
  PID=$!
  sleep 6
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.6; done
  wait $PID
  python3 -c "import json; d=json.loads(open('gpurun_out/smi_$mode.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
