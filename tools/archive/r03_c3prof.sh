#!/bin/bash
mkdir -p gpurun_out/r03_c3prof
V=nerffaceediting_amd/csrc/build/variants
(NFE_RENDER_LIB=$V/c3prof.so python3 tools/c3_profile.py bf16 8 | head -2; NFE_RENDER_LIB=$V/c3prof.so NFE_C3_LC=1 python3 tools/c3_profile.py bf16 8 | head -2) 2>&1 | grep -v "^/opt" | cut -c1-400 | tee gpurun_out/r03_c3prof/profile_lc.txt
