#!/usr/bin/env python
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv files) per kernel: mean of each counter."""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r.get("Kernel_Name", ""))[:90]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "mfx" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
