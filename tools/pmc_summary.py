#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "hc::" not in k: continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
