#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csv: mean counter value per kernel."""
import csv, glob, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[1:] or v
        print("   %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
