#!/bin/bash
# conv3_kernel ablations (timing only, wrong results): stage times of 8 views bf16 and 4 views bf16x3 per variant
V=nerffaceediting_amd/csrc/build/variants
mkdir -p gpurun_out/r03_ablate
for name in shipped abl1 abl2 abl3 abl4 shipped; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-170
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-170
done 2>&1 | tee gpurun_out/r03_ablate/ablate.log
