#!/bin/bash
# Round 4: A/B of nfe_render_bwd.hip variants (built beforehand by tools/build_variant.sh <name> nfe_render_bwd ...) on one box:
# per-kernel durations under rocprofv3 --kernel-trace --stats of `BOTH_ONLY=1 tools/time_backward.py 4 128 48 48 256`.
#   tools/r04_bwd_ab.sh <outdir under gpurun_out> base v1 v2 ...        ("base" = the shipped library)
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  if [ $v = base ]; then unset NFE_RENDER_LIB; else export NFE_RENDER_LIB=$PWD/nerffaceediting_amd/csrc/build/variants/$v.so; fi
  BOTH_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/$v.log 2>&1
  echo "== $v: $(grep backward $OUT/$v.log)"
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "bwd_" in n or "color_dot" in n:
        print(f"   {n[:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  rm -rf $OUT/$v
done
