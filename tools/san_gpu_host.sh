#!/bin/bash
# libmirt + the two C++ apps with AddressSanitizer / UBSan on the HOST code only (device code untouched), then the apps on the
# GPU: heap misuse in capi.hip / group.hip / build.hip host paths.   build: tools/san_gpu_host.sh build   run (GPU): ... run
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
S=$ROOT/raytracer_amd/san
if [ "$1" = build ]; then
  mkdir -p $S/o; cd $ROOT/raytracer_amd/csrc
  F="--offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
  for f in capi kernels build group; do /opt/rocm/bin/hipcc $F -c -o $S/o/$f.o $f.hip & done
  for f in host_bvh host_film; do /opt/rocm/lib/llvm/bin/clang++ -O2 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -c -o $S/o/$f.o $f.cpp & done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=address -fsanitize=undefined -o $S/libmirt.so $S/o/*.o -ldl -lpthread
  for a in render:render_app app:particle_app; do
    /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I $ROOT/include -o $S/${a#*:} $ROOT/apps/${a%%:*}.cpp -L $S -lmirt -Wl,-rpath,'$ORIGIN'
  done
  ls -la $S
else
  # (the ROCm ASan runtime trips a CHECK of its own inside the HSA runtime's exit handlers - "dev_runtime_unloaded" - after
  # main has returned: exit codes are not used; the logs are searched for reports instead)
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
  T=$(mktemp -d)
  run() { timeout -k 10 300 "$@" > $T/log 2>&1 || true; cat $T/log >> $T/all; grep -E "Image saved|frames/s" $T/log | tail -1 || { echo "no result from: $*"; head -40 $T/log; }; }
  run $S/render_app $T/a.bmp --w 96 --h 54 --spp 4 --depth 5
  run $S/render_app $T/b.bmp --w 96 --h 54 --spp 4 --depth 5 --parts 3 --partition tile
  run $S/render_app $T/c.bmp --w 96 --h 54 --spp 4 --depth 5 --rng 1 --parts 2 --partition spp
  run $S/particle_app --frames 2 --particles 20000 --w 160 --h 120 --spp 4 --prefix $T/p_
  run $S/particle_app --frames 2 --particles 20000 --w 160 --h 120 --spp 4 --prefix $T/q_ --host-build
  run $S/particle_app --frames 2 --particles 20000 --w 160 --h 120 --spp 16 --prefix $T/r_ --fast
  cmp $T/a.bmp $T/b.bmp || { echo "films differ"; exit 1; }
  cmp $T/p_1.bmp $T/q_1.bmp || { echo "particle films differ"; exit 1; }
  if grep -E "ERROR: AddressSanitizer|runtime error:" $T/all; then grep -E -A12 "ERROR: AddressSanitizer|runtime error:" $T/all | head -60; exit 1; fi
  echo "host sanitizers on the GPU path: no reports"
fi
