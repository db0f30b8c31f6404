# rocprofv3 PMC passes over a short bench.py run; prints the counters of kernels matching $1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcb; rm -rf $O; mkdir -p $O
ARGS="$R/bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p1 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES -d $O/p2 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL -d $O/p3 -- python3 $ARGS > /dev/null 2>&1
for d in p1 p2 p3; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f | grep -A9 "^.*$1" | grep -v "^--" ; done
find $O -name "*.db" -delete
