#!/bin/bash
# tools/shadow_blocks.sh: workgroups per CU of the shadow launches (MIRT_SHADOW_BLOCKS) under the share pool, per workload and spp
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/shadow_blocks.txt
: > $OUT
for cfg in "--spp 64" "--spp 32" "--spp 16" "--spp 8" "--workload bunny* --spp 64" "--workload buddha* --spp 64" "--workload sportsCar* --spp 64"; do
  for rep in 1 2; do
    for sb in 3 4 5; do
      MIRT_SHADOW_BLOCKS=$sb timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 $cfg > /tmp/sb.json 2> /tmp/sb.err || { echo "FAILED $cfg $sb" | tee -a $OUT; tail -3 /tmp/sb.err; exit 1; }
      python3 -c "
import json;d=json.load(open('/tmp/sb.json'))
print('%-34s shadow blocks %d: %8.2f ms %6.0f Mrays/s' % ('$cfg', $sb, d['ms_per_step'], d['value']))" | tee -a $OUT
    done
  done
done
