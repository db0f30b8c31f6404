# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box (run through gpurun):
#   kernel trace (--stats), HBM FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md prescribes), SQ counter
#   passes of the matrix-core kernels, and the bench lines themselves.  Summaries land in gpurun_out/prof/ (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py > $O/bench_blind.json 2> $O/bench_blind.err
python3 bench.py --mode nonblind --no-other-configs > $O/bench_nonblind.json 2> $O/bench_nonblind.err
python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $O/bench_blind_driver_style_20_steps.json 2>/dev/null
python3 bench.py --conv vector --no-cpu-baseline --no-other-configs > $O/bench_blind_vector.json 2>/dev/null
ICS_FUSED_GRADK=0 python3 bench.py --no-cpu-baseline --no-other-configs > $O/bench_blind_two_kernel_gradk.json 2>/dev/null
cd /tmp
ARGS="$R/bench.py --no-cpu-baseline --no-other-configs --no-other-mode"
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $ARGS > /dev/null 2>&1
SHORT="$ARGS --steps 5 --warmup 0 --no-profile --no-sustained"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq1 -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVES -d $O/sq2 -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL -d $O/sq3 -- python3 $SHORT > /dev/null 2>&1
# the two-kernel gradient path for the traffic comparison
ICS_FUSED_GRADK=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch2k -- python3 $SHORT > /dev/null 2>&1
for d in kt fetch write sq1 sq2 sq3 fetch2k; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/$d.txt 2>&1; done
python3 $R/scripts/make_traffic_json.py $O/fetch.txt $O/write.txt $O/fetch2k.txt --size 4096 --psf 15 > $O/hbm_traffic.json
python3 $R/scripts/make_mfma_json.py $O/sq2.txt $O/sq3.txt --size 4096 --psf 15 > $O/mfma_counters.json
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
tail -1 $O/bench_blind.json | cut -c1-300; head -12 $O/kt.txt; cat $O/hbm_traffic.json | head -60
