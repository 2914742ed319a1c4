#!/bin/bash
# usage: tools/sweep_env.sh "A=1 B=2 -- --streams 8" "A=2" ...   -> full bench frame per environment setting (+ bench flags after --)
for e in "$@"; do
  envs="${e%%--*}"; flags=""; case "$e" in *--*) flags="${e#*--}";; esac
  env $envs timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 $flags > /tmp/sweep.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('/tmp/sweep.json'));print('$e', round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s')"
done
