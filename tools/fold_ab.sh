#!/bin/bash
# tools/fold_ab.sh: the fold of the binary SAH tree to 8-wide nodes, greedy (MIRT_FOLD_DP=0) against the dynamic programme (1, default)
# on the four BASELINE workloads at their own sizes -> gpurun_out/fold_ab.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/fold_ab.txt
: > $OUT
run() { # workload dp args...
  wl="$1"; export MIRT_FOLD_DP=$2; shift 2
  timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --workload "$wl" "$@" > /tmp/l.json 2>/tmp/l.err || { tail -3 /tmp/l.err >> $OUT; exit 1; }
  python3 -c "
import json;d=json.load(open('/tmp/l.json'));r=d['roofline']
print('$wl fold_dp $MIRT_FOLD_DP: %.2f ms  node visits %.2f  prim tests %.2f per closest ray, own tree %.0f ms' % (d['ms_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray'], d['config']['build_ms']['own_tree_ms']))" >> $OUT
}
for v in 0 1 0 1; do run "conference*" $v --steps 5 --warmup 1 || exit 1; done
for v in 0 1 0 1; do run "bunny*" $v --steps 5 --warmup 1 || exit 1; done
for v in 0 1 0 1; do run "buddha*" $v --spp 256 --steps 3 --warmup 1 || exit 1; done
for v in 0 1; do run "sportsCar*" $v --width 3840 --height 2160 --spp 1024 --steps 1 --warmup 0 || exit 1; done
