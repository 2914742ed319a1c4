#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>"  ->  raytracer_amd/variants/libmirt_NAME.so  (use with MIRT_LIB=...)
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../variants build/v_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt $2"
/opt/rocm/bin/hipcc $F -c -o build/v_$1/kernels.o kernels.hip &
/opt/rocm/bin/hipcc $F -c -o build/v_$1/capi.o capi.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libmirt_$1.so build/v_$1/capi.o build/v_$1/kernels.o build/build.o build/group.o build/host_bvh.o build/host_film.o
echo built ../variants/libmirt_$1.so
