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

# Counter record of the dominant kernel for bench.py's roofline block (per launch; see bench.py issue_model())
import json
pat = os.environ.get("PMC_KERNEL")             # substring of the kernel name to report (default: the kernel with most GPU time)
cands = [k for k in vals if (pat in k)] if pat else list(vals)
dom0 = max(cands or list(vals), key=lambda k: sum(dur.get(k, [0])))
rec = {"kernel": dom0, "avg_ns_profiled": sum(dur[dom0]) / max(1, len(dur[dom0])), "dispatches": len(dur[dom0])}
for c, v in vals[dom0].items():
    rec[c] = sum(v) / len(v)
with open(os.path.join(root, "issue_floor.json"), "w") as f:
    json.dump(rec, f, indent=1, sort_keys=True)
print("issue_floor.json:", rec)

# HBM traffic of the dominant kernel for bench.py's roofline.traffic
import json
dom = dom0
v = vals[dom]
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024.0
    write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024.0
    out = {"kernel": dom, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
           "hbm_bytes_per_launch": 2.0 * fetch + write,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); per dispatch"}
    with open(os.path.join(root, "traffic.json"), "w") as f:
        json.dump(out, f)
    print("traffic.json:", out)
