# two SQ counter passes over a short bench run; summaries in gpurun_out/sq/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq; rm -rf $O; mkdir -p $O
SHORT="$R/bench.py --no-cpu-baseline --no-other-configs --no-other-mode --no-sustained --steps 5 --warmup 0 --no-profile"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq1 -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES -d $O/sq2 -- python3 $SHORT > /dev/null 2>&1
for d in sq1 sq2; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
grep -A9 "k_conv_mfma<15, [01]\|k_synth_gradk<15\|k_update_rows<0>" $O/sq1.txt | grep -v "^--" | head -60; grep -A9 "k_conv_mfma<15, [01]\|k_synth_gradk<15\|k_update_rows<0>" $O/sq2.txt | grep -v "^--" | head -60
