#!/bin/bash
# Round 6, late A/Bs on ONE box (kernel traces only):
#   1. decoder-backward kernel: build/variants/base.so (the library BEFORE the experiment, built with tools/build_variant.sh; not in the tree) against the shipped library (e^p shared between the appearance
#      head's softplus and its derivative), three interleaved repetitions of tools/time_backward.py 4 128 48 48 256
#   2. config 5, sample order: tools/cfg5_order.py merged | ideal
export TMPDIR=/tmp
OUT=gpurun_out/r06_late
mkdir -p $OUT
avg() { python3 - "$1" "$2" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in sys.argv[2].split("|")):
            print("   ", r["Name"][:80], "calls", r["Calls"], "avg_ns", r["AverageNs"])
PY
}
if [ -f nerffaceediting_amd/csrc/build/variants/base.so ]; then
for rep in 1 2 3; do
  for lib in nerffaceediting_amd/csrc/build/variants/base.so nerffaceediting_amd/libnfe_render.so; do
    rm -rf $OUT/st
    NFE_RENDER_LIB=$PWD/$lib BOTH_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/bwd.log 2>&1
    echo "$lib $(avg $OUT/st bwd_decoder_kernel)" | tee -a $OUT/bwd_ab.txt
  done
done
fi
for mode in merged ideal merged ideal; do
  rm -rf $OUT/st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 tools/cfg5_order.py $mode 10 > $OUT/order_$mode.log 2>&1
  tail -1 $OUT/order_$mode.log | tee -a $OUT/order_ab.txt
  avg $OUT/st "render_ws_kernel|importance_kernel" | tee -a $OUT/order_ab.txt
done
rm -rf $OUT/st
