#!/bin/bash
# tools/pool_ab.sh: share pool (one-tile home groups) against plain sample streams, alternating runs on one box, for the
# whole frame and the shares one rank renders at N = 2, 4, 8 -> gpurun_out/pool_ab.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/pool_ab.txt
: > $OUT
run() {  # label spp extra-args
  timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --spp $2 $3 > /tmp/pool_b.json 2> /tmp/pool_b.err || { echo "FAILED $*" >> $OUT; tail -3 /tmp/pool_b.err >> $OUT; return 1; }
  python3 - "$1" "$2" "$3" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/pool_b.json'))
print(f"{sys.argv[1]:<8} spp {sys.argv[2]:>3} {sys.argv[3]:<28}: {d['ms_per_step']:8.2f} ms  {d['value']:8.0f} Mrays/s  closest launches/frame {d['roofline']['launches'] / d['steps']:.0f}")
PY
  tail -1 $OUT
}
for rep in 1 2 3; do
  for spp in ${@:-64 32 16 8}; do
    run static $spp "--shares 1" && run pool $spp "--shares 0" || exit 1
  done
done
