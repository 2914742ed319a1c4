#!/bin/bash
# kernel stats of the particle app (the reference's apps/app.cu regime): tools/app_stats.sh [particles] [frames]
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/app_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- $ROOT/raytracer_amd/particle_app --frames ${2:-2} --particles ${1:-500000} --prefix $OUT/f_ $APP_ARGS > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
rm -f $OUT/kt_kernel_trace.csv $OUT/f_*.bmp
tail -2 $OUT/log
cut -c1-140 $OUT/kt_kernel_stats.csv | head -10
