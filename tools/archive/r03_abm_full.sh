#!/bin/bash
# the same ablations inside the pipeline: stage times of config 3 (8 views bf16) and the FFHQ configuration (4 views split-bf16)
V=nerffaceediting_amd/csrc/build/variants
for name in ${VARIANTS:-shipped abm128 abm130 abm138}; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-150
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-150
done
