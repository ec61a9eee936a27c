#!/bin/bash
# A/B on one box: the library of the last commit (variants/head.so) against the working tree's, stage times of both configurations
V=nerffaceediting_amd/csrc/build/variants
for rep in 1 2; do for name in head tree; do
  lib=$V/head.so; [ $name = tree ] && lib=""
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-140
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-140
done; done
