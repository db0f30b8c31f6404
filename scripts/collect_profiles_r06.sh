# rocprofv3 evidence for ONE bench.py configuration (run through gpurun; the commit travels in ICS_COMMIT, the box has no .git):
#   ICS_COMMIT=$(git rev-parse --short HEAD) scripts/collect_profiles_r06.sh SIZE PSF [MODE] [extra bench.py flags]
#   -> gpurun_out/prof_SIZE_PSF/{bench.json,kt,fetch,write,sq1,sq2,sq3}.txt, hbm_traffic.json, mfma_counters.json
# kernel trace (--stats), FETCH_SIZE / WRITE_SIZE in separate PMC runs (MI355X_MICROARCH.md), three SQ passes.
SIZE=${1:-4096}; PSF=${2:-15}; MODE=${3:-blind}; shift 3 2>/dev/null; EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${SIZE}_${PSF}
rm -rf $O; mkdir -p $O
cd $R
CFG="--size $SIZE --psf $PSF --mode $MODE --no-cpu-baseline --no-other-configs --no-other-mode $EXTRA"
python3 bench.py $CFG --steps 40 --warmup 10 > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/bench.py $CFG --steps 40 --warmup 10 --no-sustained > /dev/null 2>&1
SHORT="$R/bench.py $CFG --steps 5 --warmup 0 --no-profile --no-sustained"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq1 -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVES -d $O/sq2 -- python3 $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL -d $O/sq3 -- python3 $SHORT > /dev/null 2>&1
for d in kt fetch write sq1 sq2 sq3; do f=$(find $O/$d -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/$d.txt 2>&1; done
python3 $R/scripts/make_traffic_json.py $O/fetch.txt $O/write.txt --size $SIZE --psf $PSF > $O/hbm_traffic.json
python3 $R/scripts/make_mfma_json.py $O/sq2.txt $O/sq3.txt --size $SIZE --psf $PSF > $O/mfma_counters.json
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
echo "commit $ICS_COMMIT" > $O/COMMIT
tail -1 $O/bench.json | cut -c1-400; head -14 $O/kt.txt; head -40 $O/hbm_traffic.json
