#!/bin/bash
# tools/pool_sweep.sh [spp ...]: share pool (bench.py --shares) against plain sample streams on the bench workload -> gpurun_out/pool_sweep.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/pool_sweep.txt
: > $OUT
run() {  # spp streams shares
  timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --spp $1 --streams $2 --shares $3 > /tmp/pool_b.json 2> /tmp/pool_b.err || { echo "FAILED spp $1 streams $2 shares $3" >> $OUT; tail -3 /tmp/pool_b.err >> $OUT; return 1; }
  python3 - "$1" "$2" "$3" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/pool_b.json'))
print(f"spp {sys.argv[1]:>3} streams {sys.argv[2]:>2} shares {sys.argv[3]:>2}: {d['ms_per_step']:8.2f} ms  {d['value']:8.0f} Mrays/s  closest launches/frame {d['roofline']['launches'] / d['steps']:.0f}  avg {d['roofline']['avg_launch_ms']:.3f} ms")
PY
  tail -1 $OUT
}
for spp in ${@:-64 8}; do
  unset MIRT_POOL_GROUP
  if [ "$spp" -ge 32 ]; then run $spp 8 1 || exit 1; else run $spp 4 1 || exit 1; fi
  for g in ${POOL_GROUPS:-64 256 1024 4096 16384}; do
    export MIRT_POOL_GROUP=$g
    echo "-- MIRT_POOL_GROUP=$g" | tee -a $OUT
    if [ "$spp" -ge 32 ]; then
      run $spp 8 0 && run $spp 8 32 && run $spp 8 16 && run $spp 4 0 && run $spp 12 0 || exit 1
    else
      run $spp 4 0 && run $spp 2 0 && run $spp 6 0 || exit 1
    fi
  done
done
