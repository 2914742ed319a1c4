#!/bin/bash
# the 8-spp share (one rank of an N = 8 frame) against sample streams and pipes
for pipes in 2 4; do for st in 4 8; do
  MIRT_PIPES=$pipes timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --spp 8 --streams $st > /tmp/s.json 2>/tmp/s.err || { tail -3 /tmp/s.err; exit 1; }
  python3 -c "import json;d=json.load(open('/tmp/s.json'));print('pipes $pipes streams $st', round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"
done; done
