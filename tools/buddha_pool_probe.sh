#!/bin/bash
# tools/buddha_pool_probe.sh: share pool settings on the 256-spp buddha* workload (where the pool does not pay) -> stdout
for cfg in "--shares 1" "--shares 0" "--shares 32" "--shares 16" "--shares 128" "--shares 256" "--shares 0 --streams 4" "--shares 0 --streams 12"; do
  for sb in 3 4; do
    MIRT_SHADOW_BLOCKS=$sb timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload "buddha*" --spp 256 --steps 2 --warmup 1 $cfg > /tmp/x.json 2>/tmp/x.err || { tail -3 /tmp/x.err; exit 1; }
    python3 -c "
import json;d=json.load(open('/tmp/x.json'));k=d['roofline']['kernels']
print('%-28s sb %d: %8.2f ms | launches %d | closest %5.0f any %5.0f logic %5.0f us' % ('$cfg', $sb, d['ms_per_step'], k[0]['launches'], k[0]['avg_launch_ms']*1e3, k[1]['avg_launch_ms']*1e3, k[2]['avg_launch_ms']*1e3))"
  done
done
