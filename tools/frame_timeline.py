#!/usr/bin/env python3
"""Timeline of the LAST frame in a rocprofv3 kernel trace (frames end with k_tonemap): every kernel with its start relative
to the frame's first kernel, duration and queue, plus the GPU-idle gaps (no kernel running) longer than 5 us.

    python3 tools/frame_timeline.py <..._kernel_trace.csv>
"""
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"]
        kind = ("logicMF" if "k_logic<true" in k else "logic" if "k_logic" in k else "closest" if "k_trace_fast<false" in k else
                "shadow" if "k_trace_fast<true" in k else "redoC" if "k_redo<false" in k else "redoS" if "k_redo<true" in k else
                "fill" if "fillBuffer" in k else "copy" if "copyBuffer" in k else k.split("(")[0].split("::")[-1][:18])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id")))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2] == "k_tonemap"]
last = ends[-1]
prev = ends[-2] if len(ends) > 1 else -1
frame = rows[prev + 1:last + 1]
t0 = frame[0][0]
print(f"frame: {len(frame)} kernels, {(frame[-1][1] - t0) / 1e3:.1f} us from first kernel start to tonemap end")
busy_end = t0
idle = 0
for a, b, kind, q in frame:
    if a > busy_end:
        if a - busy_end > 5000:
            print(f"      -- GPU idle {(a - busy_end) / 1e3:.1f} us")
        idle += a - busy_end
    busy_end = max(busy_end, b)
    print(f"  t={(a - t0) / 1e3:9.1f}  q{q:>2} {kind:12s} {(b - a) / 1e3:8.1f} us")
print(f"idle (no kernel running) in frame: {idle / 1e3:.1f} us")
