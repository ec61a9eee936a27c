#!/bin/bash
# decoder-backward kernel time of library variants (rocprofv3 kernel trace of tools/time_backward.py):  tools/r05_bwd_abl.sh name[=variant.so] ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  name=${v%%=*}; lib=${v#*=}
  if [ "$lib" = "$name" ]; then lib=nerffaceediting_amd/libnfe_render.so; fi
  d=$R/gpurun_out/r05_bwd_abl_$name; rm -rf $d
  NFE_RENDER_LIB=$R/$lib BOTH_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/time_backward.py 4 128 48 48 ${PLANE:-256} > $d.log 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bwd_decoder" in r["Name"] or "scatter_sorted" in r["Name"]:
        print(f'{sys.argv[2]:12s} {r["Name"][:50]:50s} avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
done
