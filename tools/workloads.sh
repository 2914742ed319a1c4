#!/bin/bash
# every BASELINE workload stand-in at 1080p, 16 spp: ms/frame and Mrays/s
for wl in "bunny*" "buddha*" "conference*" "sportsCar*"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 2 --spp 16 --workload "$wl" > /tmp/wl.json 2>/tmp/wl.err || { tail -5 /tmp/wl.err; exit 1; }
  python3 -c "import json;d=json.load(open('/tmp/wl.json'));print(d['config']['workload'][:60], round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s', 'build', round(d['config']['scene_build_s'],2),'s', 'nodes/ray', round(d['roofline']['node_visits_per_ray'],2), 'tris/ray', round(d['roofline']['tri_tests_per_ray'],2))"
done
