#!/bin/bash
# round 6: pure kernel times (rocprofv3 kernel trace) of tools/time_up.py for variants of the up-sampling layers:
#   tools/r06_up_trace.sh <math> <views> name[:ENV=V[,ENV=V...]][=variant.so] ...
# e.g. tools/r06_up_trace.sh bf16 8 tiles:NFE_UP_STRIP=0 strips strips4:NFE_UP_STRIP_SEGS=4
export TMPDIR=/tmp
math=$1; views=$2; shift 2
mkdir -p gpurun_out/r06_dense
for v in "$@"; do
  spec=${v%%=*.so}; lib=""
  case "$v" in *=*.so) lib=${v##*=};; esac
  name=${spec%%:*}; envs=""
  case "$spec" in *:*) envs=${spec#*:};; esac
  [ -z "$lib" ] && lib=nerffaceediting_amd/libnfe_render.so
  rm -rf gpurun_out/r06_dense/tr_$name
  (
    export NFE_RENDER_LIB=$PWD/$lib
    IFS=','; for e in $envs; do export "$e"; done; unset IFS
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_dense/tr_$name -- python3 tools/time_up.py $math $views > gpurun_out/r06_dense/out_$name.txt 2>&1
  )
  python3 - "$name" "$math" "$views" <<'PY'
import csv, glob, sys
from collections import defaultdict
name = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/r06_dense/tr_{name}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "conv3_kernel" in k or "modsplit" in k or "upfir" in k or "upconv" in k:
        agg[(k.replace("void nfe::", "")[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("==", name, sys.argv[2], "x" + sys.argv[3])
for k, v in agg.items():
    v = sorted(v)
    print(f"   {k[0]:40s} grid {int(k[1])//256:5d}x{k[2]}x{k[3]}  n={len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
  rm -rf gpurun_out/r06_dense/tr_$name
done
