#!/bin/bash
V=nerffaceediting_amd/csrc/build/variants
mkdir -p gpurun_out/r03_lane
for name in "$@"; do
  echo "== $name"
  lib=$V/$name.so; [ "$name" = shipped ] && lib=""
  NFE_RENDER_LIB=$lib python3 tools/repro_lane_mask.py 60 2>&1 | grep -E "geo|REPRODUCED|repeatable|Error|error" | awk '/geo/{ if ($0 !~ / 0 entries/) {n++; if (n<=3) print}; next} {print}'
done 2>&1 | tee -a gpurun_out/r03_lane/log2.txt
