#!/usr/bin/env python3
"""dev helper: per-launch averages of a rocprofv3 --pmc counter_collection.csv, one line per kernel (12 busiest)

    python tools/sq_summary.py <dir with */*counter_collection.csv> [sort counter]
"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
key = sys.argv[2] if len(sys.argv) > 2 else "SQ_BUSY_CYCLES"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k][key])[:12]:
    launches = max(n[k].values())
    print(k, "launches", launches, {c: round(v / max(launches, 1)) for c, v in sorted(acc[k].items())})
