#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 kernel trace with start offsets, durations and gaps (us)."""
import csv, sys
rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mirt::", "").replace("mirt::", "")[:40], r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = rows[0][0]
prev_end = t0
for a, b, k, q in rows:
    print(f"{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:7.1f}  gap {(a - prev_end) / 1e3:7.1f}  q{q}  {k}")
    prev_end = max(prev_end, b)
