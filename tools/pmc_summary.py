#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch of kernels whose name contains a pattern.
usage: pmc_summary.py <counter_collection.csv> [pattern]"""
import collections
import csv
import sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "iqgpu"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if pat not in k:
        continue
    k = k.split("(")[0][-60:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k in agg:
    n = len(disp[k])
    print("%s  dispatches=%d  VGPR=?" % (k, n))
    for c, v in sorted(agg[k].items()):
        print("    %-28s %16.1f" % (c, v / n))
