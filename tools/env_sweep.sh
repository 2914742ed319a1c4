#!/bin/bash
# tools/env_sweep.sh "VAR=a VAR2=b" "VAR=c" ...   -- one bench line per environment setting (default command + $BENCH_ARGS, 4 timed frames)
for e in "$@"; do
  env $e timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 $BENCH_ARGS > /tmp/s.json 2>/tmp/s.err || { echo "$e failed"; tail -3 /tmp/s.err; continue; }
  python3 -c "
import json;d=json.load(open('/tmp/s.json'));k=d['roofline']['kernels']
print('%-44s %7.2f ms %6.0f Mrays/s | closest %5.0f any %5.0f logic %5.0f us' % ('$e', d['ms_per_step'], d['value'], k[0]['avg_launch_ms']*1e3, k[1]['avg_launch_ms']*1e3, k[2]['avg_launch_ms']*1e3))"
done
