#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (profiles/pmc_passes.sh): per kernel, mean of each
counter over its dispatches.  usage: pmc_summary.py <dir> [kernel substring ...]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{d}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ghmm::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if want and not any(w in k for w in want):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")
