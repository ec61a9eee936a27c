#!/bin/bash
# conv3_kernel: share of a wave's life spent in the epilogue (light profile build: three s_memtime stamps per wave)
mkdir -p gpurun_out/r03_c3prof
V=nerffaceediting_amd/csrc/build/variants
for m in bf16 bf16x3; do NFE_RENDER_LIB=$V/c3prof2.so python3 tools/c3_profile.py $m 8 2>&1 | grep -v "^/opt" | cut -c1-30,60-75,95-330; done | tee gpurun_out/r03_c3prof/profile_light.txt
