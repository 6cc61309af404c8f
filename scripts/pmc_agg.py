#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files under a directory: per kernel (short name) and counter, the mean per launch."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void sadvio::", "").replace("sadvio::", "")
        a = acc[(name, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:32s} {c:24s} launches {n:6d} mean {v / n:16.1f}")
