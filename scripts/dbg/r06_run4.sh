timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t4_all.txt 2>&1
echo rc=$? >> gpurun_out/t4_all.txt
