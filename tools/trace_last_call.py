#!/usr/bin/env python3
"""Kernel timeline of the LAST synthesis call in a rocprofv3 kernel trace (start marker: the last fc_grouped_kernel that is
followed by conv kernels).   python tools/trace_last_call.py <trace dir> [n_kernels]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = t0
tot = 0.0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    prev_end = e
    tot += d
    name = r["Kernel_Name"].replace("void nfe::", "").replace("nfe::", "")[:44]
    wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 256
    print(f'{(s - t0) / 1e3:9.1f} us  +{gap:6.1f}  {name:44s} grid {int(r["Grid_Size_X"]) // max(wg, 1):6d}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} wg {wg:4d} lds {r.get("LDS_Block_Size", "?"):>6s}  {d:8.1f} us')
print(f"kernel time {tot:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
