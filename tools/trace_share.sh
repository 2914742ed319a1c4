#!/bin/bash
# kernel trace of an 8-spp share frame -> gpurun_out/share8_timeline.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/share8; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --spp ${1:-8} > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd $ROOT
python3 tools/frame_timeline.py $OUT/kt_kernel_trace.csv > gpurun_out/share${1:-8}_timeline.txt
rm -f $OUT/kt_kernel_trace.csv
tail -1 gpurun_out/share${1:-8}_timeline.txt; cat $OUT/bench.json | cut -c1-200
