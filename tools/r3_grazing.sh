#!/bin/bash
# Round 3: the known class of exceptions of traversal 0, looked for where it lives -- rays almost in the plane of a triangle
# (tests/gpu_common.grazing_rays) -- per distance-pruning slack (variant libraries of tools/build_variant.sh), with the frame
# cost of each slack beside it.   tools/r3_grazing.sh [rays_per_seed [seeds]]
out=gpurun_out/r3_grazing
mkdir -p $out; : > $out/soak.txt
for v in default slack1e4 slack3e4 slack5e4 slack1e3; do
  lib=raytracer_amd/libmirt.so; [ $v != default ] && lib=raytracer_amd/variants/libmirt_$v.so
  echo "== grazing soak, $v" | tee -a $out/soak.txt
  MIRT_LIB=$lib timeout -k 10 600 python3 -u tools/soak_aimed.py ${1:-1000000} ${2:-2} grazing 2>&1 | tee -a $out/soak.txt
  [ ${PIPESTATUS[0]} -gt 1 ] && exit 1
done
tools/ab.sh $out/ab "--steps 4 --warmup 1" raytracer_amd/libmirt.so raytracer_amd/variants/libmirt_slack3e4.so raytracer_amd/variants/libmirt_slack5e4.so raytracer_amd/variants/libmirt_slack1e3.so | tee -a $out/soak.txt
