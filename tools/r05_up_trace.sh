#!/bin/bash
# round 5: pure kernel times (rocprofv3 kernel trace) of tools/time_up.py for library variants:  tools/r05_up_trace.sh name[=variant.so] ...
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_dense
for v in "$@"; do
  name=${v%%=*}; lib=${v#*=}
  if [ "$lib" = "$name" ]; then lib=nerffaceediting_amd/libnfe_render.so; fi
  rm -rf gpurun_out/r05_dense/tr_$name
  NFE_RENDER_LIB=$PWD/$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_dense/tr_$name -- python3 tools/time_up.py bf16 8 > /dev/null 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys
from collections import defaultdict
name = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/r05_dense/tr_{name}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "conv3_kernel" in k or "modsplit" in k or "upfir" in k:
        agg[(k[:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("==", name)
for k, v in agg.items():
    v = sorted(v)
    print(f"   {k[0]:44s} grid {int(k[1])//256:5d}x{k[2]}x{k[3]}  n={len(v):3d}  median {v[len(v)//2]:8.1f} us")
PY
  rm -rf gpurun_out/r05_dense/tr_$name
done
