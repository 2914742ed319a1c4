#!/bin/bash
# tools/pool_streams.sh: sample streams (slots per pixel) under the share pool, per spp -> gpurun_out/pool_streams.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/pool_streams.txt
: > $OUT
run() {  # spp streams shares
  timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --spp $1 --streams $2 --shares $3 > /tmp/pool_b.json 2> /tmp/pool_b.err || { echo "FAILED $*" >> $OUT; tail -3 /tmp/pool_b.err >> $OUT; return 1; }
  python3 - "$1" "$2" "$3" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/pool_b.json'))
print(f"spp {sys.argv[1]:>3} streams {sys.argv[2]:>2} shares {sys.argv[3]:>2}: {d['ms_per_step']:8.2f} ms  {d['value']:8.0f} Mrays/s  closest launches/frame {d['roofline']['launches'] / d['steps']:.0f}")
PY
  tail -1 $OUT
}
for s in 5 6 8 10 12 16; do run 64 $s 0 || exit 1; done
for s in 3 4 6 8 12; do run 32 $s 0 || exit 1; done
for s in 2 3 4 6 8; do run 16 $s 0 || exit 1; done
for s in 1 2 3 4 6; do run 8 $s 0 || exit 1; done
run 64 8 32; run 32 8 16; run 16 4 8; run 8 2 4
