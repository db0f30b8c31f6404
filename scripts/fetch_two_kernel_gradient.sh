cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof2k; rm -rf $O; mkdir -p $O
SHORT="$R/bench.py --no-cpu-baseline --no-other-configs --no-other-mode --steps 5 --warmup 0 --no-profile"
ICS_FUSED_GRADK=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -- python3 $SHORT > /dev/null 2>&1
f=$(find $O/f -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/fetch2k.txt 2>&1
find $O -name "*.db" -delete
grep -A1 "k_gradk_mfma<1>" $O/fetch2k.txt | tail -3
