#!/bin/bash
# the 8-, 16- and 32-spp shares (one rank of an N = 8 / 4 / 2 frame) against sample streams (two pipes)
for spp in 8 16 32; do for st in 2 4 8 16; do
  [ $st -gt $spp ] && continue
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --spp $spp --streams $st > /tmp/s.json 2>/tmp/s.err || { tail -3 /tmp/s.err; exit 1; }
  python3 -c "import json;d=json.load(open('/tmp/s.json'));print('spp $spp streams $st', round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"
done; done
