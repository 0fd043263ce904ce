#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per launch of one kernel.
usage: pmc_sum.py <kernel-substring> <dir> [<dir> ...]   (each dir = one `rocprofv3 --kernel-trace --pmc ... -d <dir>` pass)"""
import collections
import csv
import glob
import json
import os
import sys

needle, dirs = sys.argv[1], sys.argv[2:]
tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if needle not in row["Kernel_Name"]:
                    continue
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[row["Counter_Name"]].add((path, row["Dispatch_Id"]))
print(json.dumps({k: round(tot[k] / len(cnt[k])) for k in sorted(tot)}, indent=1))
print(json.dumps({k: len(cnt[k]) for k in sorted(tot)}))
