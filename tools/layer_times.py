#!/usr/bin/env python3
"""Per-layer conv timings from a rocprofv3 kernel trace of one synthesis call (run under rocprofv3)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r for r in rows if "conv_kernel" in r["Kernel_Name"] or "conv3_kernel" in r["Kernel_Name"] or "modsplit" in r["Kernel_Name"] or "torgb" in r["Kernel_Name"] or "upfir" in r["Kernel_Name"] or "render_kernel" in r["Kernel_Name"]]
for r in names[-60:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{r["Kernel_Name"][:48]:48s} grid {int(r["Grid_Size_X"])//256:5d}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {d:8.1f} us')
