#!/bin/bash
# tools/kstats.sh "<bench args>": rocprofv3 kernel stats of a bench run -> gpurun_out/kstats.csv (top rows printed)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/kstats_tmp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline $1 > $OUT/bench.json 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
cp $OUT/kt_kernel_stats.csv $ROOT/gpurun_out/kstats.csv; rm -f $OUT/kt_kernel_trace.csv
cut -c1-130 $ROOT/gpurun_out/kstats.csv | head -9
python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],'ms',d['value'],'Mrays/s')"
