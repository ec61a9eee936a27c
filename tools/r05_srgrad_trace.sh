#!/bin/bash
# per-kernel statistics of the SR-head gradient (tools/time_sr_grad.py, 4 views) under rocprofv3:  gpurun -- bash tools/r05_srgrad_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
d=$R/gpurun_out/r05_srgrad; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/time_sr_grad.py 4 > $d.log 2>&1
grep "SR head" $d.log
f=$(find $d -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.7:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
