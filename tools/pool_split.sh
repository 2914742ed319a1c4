#!/bin/bash
# tools/pool_split.sh "<bench args>" ...: per-kernel time of a frame (events around every launch), overlapped and with MIRT_SERIAL=1
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/pool_split.txt
: > $OUT
for cfg in "$@"; do
  for serial in 0 1; do
    if [ $serial = 1 ]; then export MIRT_SERIAL=1; else unset MIRT_SERIAL; fi
    timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 $cfg > /tmp/ps.json 2> /tmp/ps.err || { echo "FAILED $cfg" >> $OUT; tail -3 /tmp/ps.err >> $OUT; exit 1; }
    python3 - "$cfg" "$serial" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/ps.json'))
ks = d['roofline']['kernels']
print(f"{sys.argv[1]:<34} serial={sys.argv[2]}: {d['ms_per_step']:7.2f} ms | " + " | ".join(
    f"{k['kernel'].replace('k_trace_fast','tr')}: {k['launches']} x {k['avg_launch_ms']:.3f} = {k['launches']*k['avg_launch_ms']:.1f} ms" for k in ks))
PY
    tail -1 $OUT
  done
done
