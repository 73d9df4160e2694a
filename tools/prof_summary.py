#!/usr/bin/env python3
"""Summarise rocprofv3 csv output directories produced by tools/prof.sh."""
import csv, glob, os, sys, collections
base = sys.argv[1]
for f in sorted(glob.glob(os.path.join(base, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", f)
    print(open(f).read())
for f in sorted(glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    print("==", f)
    for k, v in agg.items():
        print(k, {a: b for a, b in v.items()})
