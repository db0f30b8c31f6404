# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box (run through gpurun):
#   kernel trace, HBM FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), and the bench lines themselves.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py > $O/bench_blind.json 2> $O/bench_blind.err
python3 bench.py --mode nonblind > $O/bench_nonblind.json 2> $O/bench_nonblind.err
python3 bench.py --conv vector --no-cpu-baseline > $O/bench_blind_vector.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode > /dev/null 2>&1
for d in kt fetch write; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
tail -1 $O/bench_blind.json | cut -c1-400; head -12 $O/kt.txt; grep -A1 "k_conv_mfma\|k_update\|k_gradk" $O/fetch.txt | head -20; grep -A1 "k_conv_mfma\|k_update\|k_gradk" $O/write.txt | head -20
