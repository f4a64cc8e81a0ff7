#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 --pmc output directory (counter_collection.csv)."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    n = len(disp[k])
    print(f"{k}  dispatches={n}")
    for c in sorted(acc[k]):
        print(f"    {c:32s} {acc[k][c] / n:18.1f}")
