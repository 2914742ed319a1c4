#!/bin/bash
# tools/iter_run.sh WORKLOAD: kernel trace of one 64-spp frame (8 streams) -> gpurun_out/iter_<name>.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/iter; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WORKLOAD="$1" timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python3 $ROOT/tools/prof_frame.py 64 1920 1080 8 > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
cd $ROOT
tail -1 $OUT/log
python3 tools/iter_profile.py $OUT/kt_kernel_trace.csv | tee gpurun_out/iter_${1%\*}.txt
rm -f $OUT/kt_kernel_trace.csv
