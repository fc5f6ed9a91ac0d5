#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (development tool)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if flt and flt not in r["Kernel_Name"]:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(f"{k}  dispatches={len(dur[k])} mean_us={sum(dur[k])/len(dur[k]):.1f}")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean={sum(v)/len(v):.5g}")
