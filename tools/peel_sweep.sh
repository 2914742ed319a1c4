#!/bin/bash
# host SAH builder with large primitives set aside (MIRT_SAH_PEEL = area fraction): frame time per workload
for wl in "conference*" "bunny*" "buddha*" "sportsCar*"; do
  for t in 0 0.25 0.5; do
    MIRT_SAH_PEEL=$t timeout -k 10 280 python3 bench.py --no-cpu-baseline --workload "$wl" --steps 2 --warmup 1 --builder host > /tmp/bc.json 2>/tmp/bc.err || { tail -5 /tmp/bc.err; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/bc.json'));c=d['config'];r=d['roofline'];print(c['workload'][:12], 'peel $t', round(d['ms_per_step'],2),'ms/frame', 'nodes/ray', round(r['node_visits_per_ray'],2), 'prims/ray', round(r['tri_tests_per_ray'],2))"
  done
done
