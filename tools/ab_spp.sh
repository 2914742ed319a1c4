#!/bin/bash
# tools/ab_spp.sh OUTDIR LIB...  -- the 64-spp frame and the 8-spp share of an N = 8 frame, per library, twice
out=$1; shift
for spp in 64 8; do echo "--- spp $spp"; tools/ab.sh $out/spp$spp "--steps 4 --warmup 1 --spp $spp" "$@" || exit 1; done
