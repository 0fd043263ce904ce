#!/usr/bin/env python
"""Top kernels of a rocprofv3 --kernel-trace --stats run: stats_top.py <dir or kernel_stats.csv> [rows]"""
import csv, glob, os, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:9.1f} us  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
print(f"total {tot / 1e6:.1f} ms of kernel time in the trace")
