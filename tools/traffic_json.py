#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summary of tools/make_profiles.sh:  traffic_json.py <pmc_summary.csv> <source name>
HBM bytes per launch of the three hot kernels = 2 x FETCH_SIZE + WRITE_SIZE (KiB), per MI355X_MICROARCH.md: separate passes,
FETCH_SIZE doubled on gfx950 (128-B requests tallied at 64 B; an upper bound for the traversal kernels' 16-B gathers)."""
import csv
import json
import sys

NAMES = {"mirt::k_trace_fast<false, false, false>": "k_trace_fast<closest>", "mirt::k_trace_fast<true, false, false>": "k_trace_fast<any>",
         "mirt::k_logic<false, false>": "k_logic"}
rows = {}
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        if r["kernel"] in NAMES and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
            rows.setdefault(NAMES[r["kernel"]], {})[r["counter"]] = (float(r["per_dispatch"]), int(r["dispatches"]))
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one pass each) -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0",
       "correction": "MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> doubled; WRITE_SIZE exact; unit KiB",
       "source": sys.argv[2],
       "configuration": {"workload": "conference*", "width": 1920, "height": 1080, "spp": 64, "depth": 8, "streams": 8, "shares": 64,
                         "traversal": 0, "n_gpus": 1},
       "kernels": {}}
for name, c in rows.items():
    f, w = c["FETCH_SIZE"], c["WRITE_SIZE"]
    out["kernels"][name] = {"launches": f[1], "FETCH_SIZE_KiB_per_launch": f[0], "WRITE_SIZE_KiB_per_launch": w[0],
                            "hbm_bytes_per_launch": (2 * f[0] + w[0]) * 1024}
json.dump(out, sys.stdout, indent=1)
