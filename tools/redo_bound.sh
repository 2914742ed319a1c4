#!/bin/bash
# bound on what a faster reference walk could buy: frames without the shadow / closest / both redo launches (WRONG results)
#   tools/build_variant.sh timing "-DMIRT_TIMING_EXPERIMENTS"   first (here, on the CPU): the shipped library ignores MIRT_SKIP_REDO
#   tools/redo_bound.sh [spp]      (BENCH_ARGS="--workload sportsCar* ..." for another configuration)
spp=${1:-64}
for v in 0 1 2 3 0; do
  MIRT_LIB=raytracer_amd/variants/libmirt_timing.so MIRT_SKIP_REDO=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps ${STEPS:-6} --warmup 1 --spp $spp $BENCH_ARGS > /tmp/s.json 2>/tmp/s.err || { tail -3 /tmp/s.err; exit 1; }
  python3 -c "import json;d=json.load(open('/tmp/s.json'));print('spp $spp skip_redo $v', round(d['ms_per_step'],2), 'ms', 'redo rays', d['config'].get('reference_walk_rays_per_frame'))"
done
