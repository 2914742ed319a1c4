#!/bin/bash
# PLOC search radius x leaf size: frame time and build time on two workloads
for wl in "conference*" "buddha*"; do
  for r in 8 16 32 64; do for leaf in 3 4; do
    MIRT_PLOC_RADIUS=$r MIRT_FAST_LEAF=$leaf timeout -k 10 280 python3 bench.py --no-cpu-baseline --workload "$wl" --steps 2 --warmup 1 --builder ploc > /tmp/bc.json 2>/tmp/bc.err || { tail -5 /tmp/bc.err; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/bc.json'));c=d['config'];r=d['roofline'];print(c['workload'][:12], 'radius $r leaf $leaf', round(d['ms_per_step'],2),'ms/frame', 'nodes/ray', round(r['node_visits_per_ray'],2), 'prims/ray', round(r['tri_tests_per_ray'],2), 'own tree', c['build_ms']['own_tree_ms'], 'ms')"
  done; done
done
