#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 kernel trace (…_kernel_trace.csv): how much of the frame has which kernels in flight."""
import csv, sys, re
from collections import defaultdict
rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"]
        kind = "logic" if "k_logic" in k else "closest" if "k_trace_fast<false" in k else "shadow" if "k_trace_fast<true" in k else \
               "redo" if "k_redo" in k else "other"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
# keep the last contiguous frame: from the last k_init onwards if present
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for a, b, k in rows:
    ev.append((a, 1, k)); ev.append((b, -1, k))
ev.sort()
live = defaultdict(int)
acc = defaultdict(int)
last = ev[0][0]
for t, d, k in ev:
    key = tuple(sorted(x for x, n in live.items() if n > 0 and x in ("logic", "closest", "shadow"))) or ("idle/other",)
    acc[key] += t - last
    last = t
    live[k] += d
tot = sum(acc.values())
print(f"span {tot / 1e6:.1f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {'+'.join(k):28s} {v / 1e6:8.2f} ms  {100 * v / tot:5.1f} %")
