#!/usr/bin/env python3
"""Average per-dispatch counter values of one kernel from a rocprofv3 --pmc CSV
(…_counter_collection.csv).  Usage: pmc_summary.py <csv> [kernel-substring]"""
import csv
import sys
from collections import defaultdict

want = sys.argv[2] if len(sys.argv) > 2 else "k_cost_v3"
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if want in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for k in sorted(acc):
    print(f"{k},{acc[k][0] / acc[k][1]:.6g},{acc[k][1]}")
