#!/usr/bin/env python3
"""Per-stream chains of a rocprofv3 kernel trace: for the iterations in the middle of the last frame, each kernel's start
(relative), duration and the gap since the previous kernel of the same stream ended.  Shows what an iteration waits for.

    python3 tools/chain.py <..._kernel_trace.csv> [first_row [n_rows]]
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"]
        kind = ("logicMF" if "k_logic<true" in k else "logic" if "k_logic" in k else "closest" if "k_trace_fast<false" in k else
                "shadow" if "k_trace_fast<true" in k else "redoC" if "k_redo<false" in k else "redoS" if "k_redo<true" in k else
                "fill" if "fillBuffer" in k else "copy" if "copyBuffer" in k else k.split("(")[0][-18:])
        sid = r.get("Stream_Id") or r.get("Queue_Id")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, sid))
rows.sort()
t0 = rows[0][0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
last_end = {}
for a, b, kind, sid in rows[:first]:
    last_end[sid] = max(last_end.get(sid, 0), b)
print(f"{len(rows)} kernels, span {(max(r[1] for r in rows) - t0) / 1e6:.2f} ms; rows {first}..{first + n}")
for a, b, kind, sid in rows[first:first + n]:
    gap = a - last_end.get(sid, a)
    print(f"  t={(a - t0) / 1e3:10.1f} us  stream {sid:>3}  {kind:10s} dur {(b - a) / 1e3:8.1f} us  gap {gap / 1e3:8.1f} us")
    last_end[sid] = max(last_end.get(sid, 0), b)
# totals per kind
tot = defaultdict(lambda: [0, 0])
for a, b, kind, sid in rows:
    tot[kind][0] += 1
    tot[kind][1] += b - a
print("kind, calls, total ms, avg us")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:12s} {c:6d} {t / 1e6:9.2f} {t / c / 1e3:9.1f}")
