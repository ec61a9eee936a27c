#!/bin/bash
mkdir -p gpurun_out/r03_c3prof
V=nerffaceediting_amd/csrc/build/variants
(NFE_RENDER_LIB=$V/c3prof.so python3 tools/c3_profile.py bf16 8; NFE_RENDER_LIB=$V/c3prof.so NFE_C3_WIDE8=1 python3 tools/c3_profile.py bf16 8 | head -3; NFE_RENDER_LIB=$V/c3prof.so python3 tools/c3_profile.py bf16x3 4) 2>&1 | grep -v "^/opt" | tee gpurun_out/r03_c3prof/profile.txt
