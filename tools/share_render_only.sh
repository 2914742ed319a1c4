#!/bin/bash
# tools/share_render_only.sh: per-rank shares of an N-rank frame priced on one GPU, with and without the film hand-over that only rank 0
# runs (bench.py --no-film) -> gpurun_out/share_render_only.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/share_render_only.txt
: > $OUT
for rep in 1 2; do
for spp in 64 32 16 8; do
  for f in "" "--no-film"; do
    timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --spp $spp $f > /tmp/s.json 2>/tmp/s.err || { tail -3 /tmp/s.err >> $OUT; exit 1; }
    python3 -c "
import json;d=json.load(open('/tmp/s.json'))
print('spp %3d %-10s %7.2f ms' % ($spp, '$f' or 'frame', d['ms_per_step']))" >> $OUT
  done
done
done
