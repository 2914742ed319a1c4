#!/bin/bash
for cfg in "2 8" "3 8" "3 9" "3 12" "3 6" "2 6" "2 10" "2 12" "2 8"; do set -- $cfg
  MIRT_PIPES=$1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --streams $2 > /tmp/s.json 2>/tmp/s.err || { tail -3 /tmp/s.err; continue; }
  python3 -c "import json;d=json.load(open('/tmp/s.json'));print('pipes $1 streams $2', round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"
done
