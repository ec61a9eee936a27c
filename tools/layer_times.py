#!/usr/bin/env python3
"""Per-layer conv timings from a rocprofv3 kernel trace of one synthesis call (run under rocprofv3)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r for r in rows if "nfe::" in r["Kernel_Name"]]
LAST = int(sys.argv[2]) if len(sys.argv) > 2 else 110
for r in names[-LAST:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{r["Kernel_Name"][:48]:48s} grid {int(r["Grid_Size_X"])//256:5d}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {d:8.1f} us')
