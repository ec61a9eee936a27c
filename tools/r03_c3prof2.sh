#!/bin/bash
# conv3_kernel phase profile (cycles per wave) of the shipped loop and of its timing-only ablations: which part makes the LDS-DMA slow?
mkdir -p gpurun_out/r03_c3prof
V=nerffaceediting_amd/csrc/build/variants
for v in c3prof c3prof_m2 c3prof_m8 c3prof_m10; do
  echo "== $v (C3_ABM: 2 no MFMA, 8 no fragment reads)"
  NFE_RENDER_LIB=$V/$v.so python3 tools/c3_profile.py bf16 8 2>&1 | grep -v "^/opt" | head -2 | cut -c1-420
done | tee gpurun_out/r03_c3prof/profile_abm.txt
