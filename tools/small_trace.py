#!/usr/bin/env python
"""Ordered kernel timeline of the LAST graphed step in a `rocprofv3 --kernel-trace` csv (tools/prof_round.sh)."""
import csv, glob, os, sys
path = sys.argv[1] if len(sys.argv) > 1 else max(glob.glob("gpurun_out/prof_small_trace/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = prev = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:5.1f} dur {(e - s) / 1e3:6.1f}  grid {r['Grid_Size_X']:>7}/{r['Workgroup_Size_X']:>4} {r['Kernel_Name'][:96]}")
    prev = e
print(f"{b - a} kernels, {(prev - t0) / 1e3:.1f} us")
