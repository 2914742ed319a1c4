#!/bin/bash
# Round profile set for bench.py: kernel-trace stats, then HBM traffic counters (one rocprofv3 run per pass).
#   tools/make_profiles.sh TAG      -> gpurun_out/prof_TAG/{kernel_stats.csv,pmc_summary.csv,bench.json}
set -e
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-x}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats.csv
rm -f $OUT/kt/kt_kernel_trace.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/p1 -o p1 --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/p1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/p2 -o p2 --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/p2.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/p1 $OUT/p2 > $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/kernel_stats.csv | cut -c1-160
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc_summary.csv | grep mirt
python3 tools/traffic_json.py $OUT/pmc_summary.csv profiles/${TAG}_pmc_fetch_write.csv > $OUT/traffic.json
cat $OUT/bench.json
