#!/bin/bash
# the builders of the traversal's own tree: scene build times and the frame rate each tree gives, BASELINE workloads at 1080p 64 spp
for wl in "bunny*" "buddha*" "conference*" "sportsCar*"; do
  for b in host device ploc; do
    timeout -k 10 280 python3 bench.py --no-cpu-baseline --workload "$wl" --steps 2 --warmup 1 --builder $b > /tmp/bc.json 2>/tmp/bc.err || { tail -5 /tmp/bc.err; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/bc.json'));c=d['config'];r=d['roofline'];print(c['workload'][:12], c['builder'], round(d['ms_per_step'],2),'ms/frame', round(d['value']),'Mrays/s', 'nodes/ray', round(r['node_visits_per_ray'],2), 'prims/ray', round(r['tri_tests_per_ray'],2), 'create', round(c['scene_build_s'],3),'s own tree', c['build_ms']['own_tree_ms'])"
  done
done
