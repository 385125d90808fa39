#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) to the library's own kernels
plus one line for everything else.  Usage: prof_summary.py <kernel_stats.csv> [title]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
print(f"# {title}")
print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,stddev_ns")
other = 0
for r in rows:
    name = r["Name"]
    if "k_" in name and "at::native" not in name:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        print(f'"{short}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["MinNs"]},{r["MaxNs"]},{float(r["StdDev"]):.0f}')
    else:
        other += int(r["TotalDurationNs"])
print(f'"(all other kernels: torch input generation, copies)",,{other},,,,')
