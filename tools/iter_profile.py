#!/usr/bin/env python3
"""How a frame's work is spread over its wavefront iterations: k_logic / traversal launch durations in launch order from a
rocprofv3 kernel trace of ONE frame (tools/prof_frame.py), as a coarse histogram over the frame.

    python3 tools/iter_profile.py <kernel_trace.csv>
"""
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"]
        kind = ("logic" if "k_logic" in k else "closest" if "k_trace_fast<false" in k else "shadow" if "k_trace_fast<true" in k else None)
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
nb = 20
bins = [dict(logic=0.0, closest=0.0, shadow=0.0, n=0) for _ in range(nb)]
for a, b, kind in rows:
    i = min(nb - 1, int((a - t0) * nb / (t1 - t0)))
    bins[i][kind] += (b - a) / 1e3
    bins[i]["n"] += kind == "logic"
print(f"frame span {(t1 - t0) / 1e6:.2f} ms, {sum(b['n'] for b in bins)} k_logic launches")
for i, b in enumerate(bins):
    print(f"  {i * 100 // nb:3d}%  logic launches {b['n']:3d}  busy us: logic {b['logic']:8.0f} closest {b['closest']:8.0f} shadow {b['shadow']:8.0f}")
