# rocprofv3 PMC passes over a stand-alone harness binary; prints the counters of the kernels matching $2
#   bash scripts/prof_pmc_tool.sh image-cases-studies_amd/csrc/tools/bench_synth_gradk k_synth_gradk
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$R/$1; PAT=$2; O=$R/gpurun_out/pmct; rm -rf $O; mkdir -p $O
export ICS_BENCH_REPS=20
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES -d $O/p2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL -d $O/p3 -- $B > /dev/null 2>&1
for d in p1 p2 p3; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/$d.txt 2>&1; echo "== $d"; awk -v pat="$PAT" '/^[^ ]/ {on = (index($0, pat) > 0 && $0 !~ /avg_us/ && NF < 6)} on' $O/$d.txt | head -12; done
find $O -name "*.db" -delete
