timeout 1700 python -m pytest tests -q -m gpu -x > gpurun_out/t9_all.txt 2>&1
echo rc=$? >> gpurun_out/t9_all.txt
python bench.py > gpurun_out/b9_default.json 2> gpurun_out/b9_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/b9_driver_style.json 2> gpurun_out/b9_driver_style.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke9.txt 2>&1
