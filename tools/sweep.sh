#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   -> ms/frame of a 16-spp bench frame per value
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --split --spp 16 --steps 2 > /tmp/sweep.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('/tmp/sweep.json'));print('$var=$v', round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s', 'one frame kernel ms', {k: round(v, 1) for k, v in d['roofline']['one_frame_kernel_ms'].items()})"
done
