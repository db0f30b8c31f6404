# rocprofv3 PMC passes over the stand-alone matrix-core convolution harness (tools/bench_conv_mfma.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$R/image-cases-studies_amd/csrc/tools/bench_conv_mfma_${1:-0}
rm -rf $R/gpurun_out/pm1 $R/gpurun_out/pm2 $R/gpurun_out/pm3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pm1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES -d $R/gpurun_out/pm2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL -d $R/gpurun_out/pm3 -- $B > /dev/null 2>&1
for d in pm1 pm2 pm3; do f=$(find $R/gpurun_out/$d -name "*.db" | head -1); echo "== $d"; python3 $R/scripts/rocprof_summary.py $f | grep -v "^ *$" | grep -A12 "k_conv_mfma<15, 0>" ; done
