#!/bin/bash
# tools/ab_lib.sh "<bench args>" VARIANT ...: the shipped library against raytracer_amd/variants/libmirt_VARIANT.so, alternating, overlapped and
# serial (MIRT_SERIAL=1: per-kernel times without company) -> gpurun_out/ab_lib.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/ab_lib.txt
: > $OUT
ARGS="$1"; shift
one() {  # label serial
  if [ "$1" = base ]; then unset MIRT_LIB; else export MIRT_LIB=$ROOT/raytracer_amd/variants/libmirt_$1.so; fi
  if [ "$2" = 1 ]; then export MIRT_SERIAL=1; else unset MIRT_SERIAL; fi
  timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > /tmp/ab.json 2> /tmp/ab.err || { echo "FAILED $1" >> $OUT; tail -3 /tmp/ab.err >> $OUT; return 1; }
  python3 - "$1" "$2" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
ks = d['roofline']['kernels']
print(f"{sys.argv[1]:<10} serial={sys.argv[2]}: {d['ms_per_step']:7.2f} ms | " + " | ".join(
    f"{k['kernel'].replace('k_trace_fast','tr')}: {k['launches']} x {k['avg_launch_ms']:.3f} = {k['launches']*k['avg_launch_ms']:.1f}" for k in ks))
PY
  tail -1 $OUT
}
for rep in 1 2; do
  for v in base "$@"; do one $v 0 || exit 1; done
done
for v in base "$@"; do one $v 1 || exit 1; done
