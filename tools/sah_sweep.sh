#!/bin/bash
# tools/sah_sweep.sh ["workload" [bench args]]: host SAH builder knobs (MIRT_SAH_BINS, MIRT_SAH_PEEL) -> gpurun_out/sah_sweep.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/sah_sweep.txt
WL="${1:-conference*}"; shift
[ -n "$APPEND" ] || : > $OUT
CFGS=${CFGS:-16:0.5 32:0.5 64:0.5 16:0.25 16:0.75 16:0 16:0.5}
for cfg in $CFGS; do
  export MIRT_SAH_BINS=${cfg%%:*} MIRT_SAH_PEEL=${cfg##*:}
  timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --steps ${STEPS:-5} --warmup ${WARMUP:-1} --workload "$WL" "$@" > /tmp/l.json 2>/tmp/l.err || { tail -3 /tmp/l.err >> $OUT; exit 1; }
  python3 -c "
import json;d=json.load(open('/tmp/l.json'));r=d['roofline']
print('$WL bins $MIRT_SAH_BINS peel $MIRT_SAH_PEEL: %.2f ms  node visits %.2f  prim tests %.2f per closest ray, build %.0f ms' % (d['ms_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray'], d['config']['build_ms']['own_tree_ms']))" >> $OUT
done
