#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one dir per pass) into per-kernel averages per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [per-dispatch values]
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per_dispatch.items():
        vals[names[d]][c].append(v)
for f in glob.glob(os.path.join(root, "*", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(vals, key=lambda k: -sum(dur.get(k, [0]))):
    d = dur.get(k, [])
    print(f"== {k[:90]}  dispatches/pass~{len(d)//max(1,len(glob.glob(os.path.join(root,'*.log'))))}  avg_ns={sum(d)/max(1,len(d)):.0f}")
    for c in sorted(vals[k]):
        v = vals[k][c]
        print(f"   {c:32s} avg/dispatch {sum(v)/len(v):.4g}   (n={len(v)})")
