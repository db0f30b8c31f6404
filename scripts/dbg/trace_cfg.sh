# kernel trace of scripts/dbg/time_config.py under one build of the library: scripts/dbg/trace_cfg.sh LIB SIZE PSF BLIND [STEPS]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trc; rm -rf $O; mkdir -p $O
export ICS_HIP_LIB=$R/$1 ICS_TC_PROFILE=0
rocprofv3 --kernel-trace -d $O/kt -- python3 $R/scripts/dbg/time_config.py $2 $3 $4 ${5:-1000} > $O/out.txt 2>&1
f=$(find $O/kt -name "*.db" | head -1); python3 $R/scripts/rocprof_summary.py $f > $O/kt.txt 2>&1
find $O -name "*.db" -delete
tail -1 $O/out.txt; head -${LINES_KT:-8} $O/kt.txt | cut -c1-130
