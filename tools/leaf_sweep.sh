#!/bin/bash
# tools/leaf_sweep.sh ["workload" [bench args]]: the own tree's leaves - largest leaf (MIRT_FAST_LEAF) x the price of dividing a would-be
# leaf (MIRT_SAH_LEAF_CT, in primitive tests; 1e30 = never divide) -> gpurun_out/leaf_sweep.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/leaf_sweep.txt
WL="${1:-conference*}"; shift
[ -n "$APPEND" ] || : > $OUT
CFGS=${CFGS:-3:1e30 2:1e30 3:0 3:0.5 3:1 4:0.5 4:1 8:1 3:1e30}
for cfg in $CFGS; do
  export MIRT_FAST_LEAF=${cfg%%:*} MIRT_SAH_LEAF_CT=${cfg##*:}
  timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --steps ${STEPS:-5} --warmup ${WARMUP:-1} --workload "$WL" "$@" > /tmp/l.json 2>/tmp/l.err || { tail -3 /tmp/l.err >> $OUT; exit 1; }
  python3 -c "
import json;d=json.load(open('/tmp/l.json'));r=d['roofline']
print('$WL leaf<=$MIRT_FAST_LEAF ct $MIRT_SAH_LEAF_CT: %.2f ms  node visits %.2f  prim tests %.2f per closest ray' % (d['ms_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray']))" >> $OUT
done
