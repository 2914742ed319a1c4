#!/bin/bash
# Round 3, VERDICT item 1: what the provably exact mode (traversal 1) costs next to the default, and how the aimed-soak
# difference count and the frame time move with the distance-pruning slack of traversal 0 (variant libraries built by
# tools/build_variant.sh slack1e4 "-DMIRT_SLACK=1.0e-4f", slack1e3 "-DMIRT_SLACK=1.0e-3f").
#   tools/r3_exact_cost.sh [soak|cost|all]
what=${1:-all}
out=gpurun_out/r3_exact
mkdir -p $out
line() { python3 -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], d['config']['workload'][:60], round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s')" "$1" "$2"; }
if [ $what != soak ]; then
  for trav in 0 1; do
    for wl in "bunny* 1920 1080 64 2" "buddha* 1920 1080 256 1" "conference* 1920 1080 64 2" "sportsCar* 3840 2160 1024 1"; do
      set -- $wl
      timeout -k 10 400 python3 bench.py --no-cpu-baseline --traversal $trav --workload "$1" --width $2 --height $3 --spp $4 --steps $5 --warmup 1 \
        > $out/wl_t${trav}_${1%\*}.json 2> $out/wl.err || { tail -5 $out/wl.err; exit 1; }
      line $out/wl_t${trav}_${1%\*}.json "traversal $trav"
    done
  done
  for v in slack1e4 slack1e3; do
    MIRT_LIB=raytracer_amd/variants/libmirt_$v.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $out/bench_$v.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
    line $out/bench_$v.json $v
  done
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $out/bench_default.json 2> $out/b.err || exit 1
  line $out/bench_default.json default
fi
if [ $what != cost ]; then
  for v in default slack1e4 slack1e3; do
    lib=raytracer_amd/libmirt.so; [ $v != default ] && lib=raytracer_amd/variants/libmirt_$v.so
    echo "== aimed soak, $v" | tee -a $out/soak.txt
    MIRT_LIB=$lib timeout -k 10 900 python3 -u tools/soak_aimed.py 1000000 4 2>&1 | tee -a $out/soak.txt
    [ ${PIPESTATUS[0]} -gt 1 ] && exit 1
  done
fi
exit 0
