#!/usr/bin/env python3
"""Per-step time of the dense kernels from rocprofv3 kernel_stats csv files:  python tools/ks_table.py file.csv [file.csv ...]"""
import csv
import sys

KEYS = ("conv3_kernel", "upfir", "torgb", "rgb_combine", "modsplit", "conv_kernel", "splitk", "resize", "nhwc_to")
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    calls = [int(r["Calls"]) for r in rows if "render_kernel<false, false" in r["Name"]]
    calls = calls[0] if calls else 1
    print("==", f)
    tot = 0.0
    for r in rows:
        if any(k in r["Name"] for k in KEYS):
            per = float(r["TotalDurationNs"]) / calls / 1e3
            tot += per
            print(f"   {r['Name'][:62]:62s} {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:8.1f} us  per step {per:8.1f} us")
    print(f"   dense kernels listed: {tot:8.1f} us per step")
