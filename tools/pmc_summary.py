#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSVs per (kernel, counter):  pmc_summary.py <dir> [<dir> ...] > summary.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

acc = defaultdict(float)
calls = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
                calls[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
print("kernel,counter,dispatches,sum,per_dispatch")
for (k, c), v in sorted(acc.items()):
    n = len(calls[(k, c)])
    print(f'"{k}",{c},{n},{v:.6g},{v / max(1, n):.6g}')
