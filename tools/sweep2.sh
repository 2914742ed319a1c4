#!/bin/bash
for args in "--workload bunny*" "--spp 8"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/env_sweep.sh "X=0" "MIRT_STATIC_16TH=8" "MIRT_REFILL_MIN=16" "MIRT_SHADOW_BLOCKS=4" "MIRT_STATIC_16TH=8 MIRT_REFILL_MIN=16" "MIRT_STATIC_16TH=8 MIRT_REFILL_MIN=16 MIRT_SHADOW_BLOCKS=4" "X=1"
done
