#!/bin/bash
# lane-mask delta debugging: each variant library through tools/repro_lane_mask.py (summary lines only)
V=nerffaceediting_amd/csrc/build/variants
mkdir -p gpurun_out/r03_lane
for name in "$@"; do
  echo "== $name"
  NFE_RENDER_LIB=$V/$name.so python3 tools/repro_lane_mask.py 10 2>&1 | grep -E "geo|REPRODUCED|repeatable|Error|error" | awk '/geo/{n++; if (n<=3) print; next} {print}'
done 2>&1 | tee -a gpurun_out/r03_lane/log.txt
