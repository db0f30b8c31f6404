# kernel trace of a short bench run; summary printed (gpurun_out/tr/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tr; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-other-mode --no-sustained --steps ${1:-100} --warmup 10 --no-profile "${@:2}" > /dev/null 2>&1
f=$(find $O/kt -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/kt.txt 2>&1
find $O -name "*.db" -delete
head -80 $O/kt.txt | cut -c1-150
