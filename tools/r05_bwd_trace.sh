#!/bin/bash
# Per-kernel statistics of the editing-size backward (4 views of 128^2 x (48+48), both plane sets), default build and the
# single-wave decoder kernel of round 4 (NFE_BWD_DECODER=single), on one box.   gpurun -- bash tools/r05_bwd_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in default single; do
  d=$R/gpurun_out/r05_bwd_$mode
  rm -rf $d
  if [ $mode = single ]; then export NFE_BWD_DECODER=single; else unset NFE_BWD_DECODER; fi
  BOTH_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/time_backward.py 4 128 48 48 256 > $d.log 2>&1
  echo "== $mode"; tail -1 $d.log
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
done
