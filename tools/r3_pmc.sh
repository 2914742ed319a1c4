#!/bin/bash
# Round 3 counter evidence for the bench configuration (conference*, 1080p, 64 spp, 8 streams, pool of 64 shares): SQ / TCC / TCP passes with each
# kernel alone, the serial kernel trace, then profiles/r3_pmc_sq.{csv,json}.   Run on the GPU box from the repo root.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tools/pmc_run.sh 64 1920 1080 8 64
cp gpurun_out/pmc_summary.csv gpurun_out/r3_pmc_sq.csv
rays=$(grep -h "closest_rays" gpurun_out/pmc/p1.log | tail -1)
export TMPDIR=/tmp
( cd /tmp && MIRT_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r3_serial -o kt --output-format csv -- python3 $OLDPWD/tools/prof_frame.py 64 1920 1080 8 64 > $OLDPWD/gpurun_out/r3_serial.log 2>&1 )
stats=$(find gpurun_out/r3_serial -name "kt_kernel_stats.csv" | head -1)
cp $stats gpurun_out/r3_serial_kernel_stats.csv
find gpurun_out/r3_serial -name "*kernel_trace.csv" -delete
python3 tools/pmc_sq_json.py gpurun_out/r3_pmc_sq.csv gpurun_out/r3_serial_kernel_stats.csv "$rays" > gpurun_out/r3_pmc_sq.json
cat gpurun_out/r3_pmc_sq.json
