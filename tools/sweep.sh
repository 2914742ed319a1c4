#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   -> ms/frame of a 16-spp bench frame per value
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --spp 16 --steps 2 > /tmp/sweep.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('/tmp/sweep.json'));print('$var=$v', round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s', 'closest', round(d['roofline']['closest_kernel_ms'],1),'shadow', round(d['roofline']['shadow_kernel_ms'],1),'logic', round(d['roofline']['logic_kernel_ms'],1))"
done
