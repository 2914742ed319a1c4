#!/usr/bin/env python3
"""profiles/r3_pmc_sq.json from the SQ / TCP counter passes of tools/pmc_run.sh (kernels serialised by the counter
collection, i.e. each running alone) and the kernel trace of the same frame run serially (MIRT_SERIAL=1: durations alone):

    pmc_sq_json.py <pmc_summary.csv> <serial kernel_stats.csv> <rays json from prof_frame's print> > profiles/r3_pmc_sq.json

Per hot kernel:  lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)  (lanes active per VALU instruction);
wait_frac / issue_stall_frac / active_frac = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (wave
parked on s_waitcnt / stalled on issue / executing; all in quad-cycles, MI355X_MICROARCH.md);  valu_issue_frac = VALU-busy
SIMD time over the SIMD time of the kernel alone = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8);
l1_req_frac = vector-L1 accesses per second (TCP_TOTAL_CACHE_ACCESSES over the kernel's serial duration) over the 530 G
lane-requests/s the chip's vector L1s serve for divergent 16-byte gathers (tools/micro/gather_rate.hip)."""
import csv
import json
import sys

NAMES = {"mirt::k_trace_fast<false, false, false>": "k_trace_fast<closest>", "mirt::k_trace_fast<true, false, false>": "k_trace_fast<any>",
         "mirt::k_logic<false, false>": "k_logic"}
L1_CEILING = 530e9
pmc = {}
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        if r["kernel"] in NAMES:
            pmc.setdefault(NAMES[r["kernel"]], {})[r["counter"]] = (float(r["sum"]), int(r["dispatches"]))
dur = {}
with open(sys.argv[2], newline="") as fh:
    for r in csv.DictReader(fh):
        k = r["Name"].split("(")[0].replace("void ", "")
        if k in NAMES:
            dur[NAMES[k]] = (float(r["TotalDurationNs"]) * 1e-9, int(r["Calls"]))
rays = json.loads(sys.argv[3].replace("'", '"')) if len(sys.argv) > 3 else {}
out = {"command": "tools/r3_pmc.sh (tools/pmc_run.sh 64 1920 1080 8 64: one rocprofv3 --pmc pass per counter set over tools/prof_frame.py; "
                  "MIRT_SERIAL=1 rocprofv3 --kernel-trace --stats over the same program for durations alone)",
       "configuration": {"workload": "conference*", "width": 1920, "height": 1080, "spp": 64, "depth": 8, "streams": 8, "shares": 64, "traversal": 0,
                         "n_gpus": 1},
       "frame": rays, "units": "SQ_* in quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs", "kernels": {}}
for name, c in pmc.items():
    g = lambda k: c.get(k, (0.0, 0))[0]
    wc = g("SQ_WAVE_CYCLES")
    k = {"dispatches": c["SQ_WAVE_CYCLES"][1] if "SQ_WAVE_CYCLES" in c else None,
         "lane_util": g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else None,
         "wait_frac": g("SQ_WAIT_ANY") / wc if wc else None,
         "issue_stall_frac": g("SQ_WAIT_INST_ANY") / wc if wc else None,
         "active_frac": g("SQ_ACTIVE_INST_ANY") / wc if wc else None,
         "valu_wave_instructions": g("SQ_INSTS_VALU"),
         "valu_issue_frac": 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * g("GRBM_GUI_ACTIVE") / 8) if g("GRBM_GUI_ACTIVE") else None,
         "l1_accesses": g("TCP_TOTAL_CACHE_ACCESSES_sum"), "l2_read_requests": g("TCP_TCC_READ_REQ_sum"),
         "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else None}
    if name in dur:
        k["serial_seconds"], k["serial_calls"] = dur[name]
        k["l1_req_frac"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / dur[name][0] / L1_CEILING if dur[name][0] else None
    out["kernels"][name] = k
json.dump(out, sys.stdout, indent=1)
