# rocprofv3 kernel trace (+ idle gaps, scripts/rocprof_summary.py) of the device-resident deblur_module run of scripts/driver_timing.py:
#   scripts/driver_trace.sh [size] [blur] [iterations]  ->  gpurun_out/driver_trace_SIZE.txt
SIZE=${1:-2048}; BW=${2:-15}; IT=${3:-20}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/driver_trace
rm -rf $O; mkdir -p $O
python3 $R/scripts/driver_timing.py $SIZE $BW $IT > $R/gpurun_out/driver_timing_$SIZE.txt 2>&1
ICS_DRIVER_ONLY_RESIDENT=1 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O -- python3 $R/scripts/driver_timing.py $SIZE $BW $IT > /dev/null 2>&1
f=$(find $O -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $R/gpurun_out/driver_trace_$SIZE.txt 2>&1
find $O -name "*.db" -delete
cat $R/gpurun_out/driver_timing_$SIZE.txt; head -60 $R/gpurun_out/driver_trace_$SIZE.txt
