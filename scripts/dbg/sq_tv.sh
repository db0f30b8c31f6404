# SQ counters of the MM-TV term kernel (non-blind 4096^2, tv_mode 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sqtv; rm -rf $O; mkdir -p $O
SHORT="$R/bench.py --no-cpu-baseline --no-other-configs --no-other-mode --no-sustained --steps 5 --warmup 0 --no-profile --mode nonblind --tv-mode 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/sq1 -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum -d $O/sq2 -- python3 $SHORT > /dev/null 2>&1
for d in sq1 sq2; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete
grep -A10 "^void (anonymous namespace)::k_tvterm<1, true>(IcsTvTermArgs, int)$" $O/sq1.txt | head -12; grep -A10 "^void (anonymous namespace)::k_tvterm<1, true>(IcsTvTermArgs, int)$" $O/sq2.txt | head -12; grep "k_tvterm<1, true>" $O/sq1.txt | head -2
