#!/bin/bash
# every BASELINE config at its own size on one GPU (cfg2..cfg5), one JSON line each into gpurun_out/workloads.jsonl
out=gpurun_out/workloads.jsonl
mkdir -p gpurun_out; : > $out
run() {  # workload width height spp steps warmup
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --workload "$1" --width $2 --height $3 --spp $4 --steps $5 --warmup $6 > /tmp/wl.json 2>/tmp/wl.err || { tail -5 /tmp/wl.err; exit 1; }
  cat /tmp/wl.json >> $out
  python3 -c "import json;d=json.load(open('/tmp/wl.json'));print(d['config']['workload'][:70], round(d['ms_per_step'],2),'ms', round(d['value']),'Mrays/s', 'build', round(d['config']['scene_build_s'],2),'s', 'frac', round(d['roofline']['frac'],3))"
}
run "bunny*" 1920 1080 64 3 1 &&
run "buddha*" 1920 1080 256 2 1 &&
run "conference*" 1920 1080 64 3 1 &&
run "sportsCar*" 3840 2160 1024 1 0
