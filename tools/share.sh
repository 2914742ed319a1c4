#!/bin/bash
# one rank's share of an N-GPU frame on one GPU: bench.py --spp 64/N (efficiency = full / N / share)
for spp in 64 32 16 8; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --spp $spp > /tmp/s.json 2>/tmp/s.err || { tail -3 /tmp/s.err; exit 1; }
  python3 -c "import json;d=json.load(open('/tmp/s.json'));print('spp $spp', round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"
done
