#!/bin/bash
# usage: tools/sweep_lib.sh name1 name2 ...  (variants under raytracer_amd/variants; "base" = the in-tree library)
for v in "$@"; do
  if [ "$v" = base ]; then unset MIRT_LIB; else export MIRT_LIB=$PWD/raytracer_amd/variants/libmirt_$v.so; fi
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --split --spp 16 --steps 2 > /tmp/sweep.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('/tmp/sweep.json'));print('$v', round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s', 'one frame kernel ms', {k: round(v, 1) for k, v in d['roofline']['one_frame_kernel_ms'].items()})"
done
