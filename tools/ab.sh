#!/bin/bash
# tools/ab.sh OUTDIR "ARGS" LIB...   -- bench.py once per library (MIRT_LIB), twice round robin, one summary line each
out=$1; args=$2; shift 2
mkdir -p $out
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    MIRT_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline $args > $out/$name.$rep.json 2> $out/$name.$rep.err || { echo "$name failed"; tail -3 $out/$name.$rep.err; exit 1; }
    python - $out/$name.$rep.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
k = r.get("kernels") or [{"avg_launch_ms": 0}] * 3
print(f'{sys.argv[2]:28s} {d["ms_per_step"]:8.2f} ms  {d["value"]:8.0f} Mrays/s  closest launch {r["avg_launch_ms"]*1e3:6.1f} us  redo rays {d["config"].get("reference_walk_rays_per_frame", -1):.0f}  | split frame: closest {k[0]["avg_launch_ms"]*1e3:5.0f} any {k[1]["avg_launch_ms"]*1e3:5.0f} logic {k[2]["avg_launch_ms"]*1e3:5.0f} us', flush=True)
PY
  done
done
