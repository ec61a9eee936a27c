#!/bin/bash
# conv3_kernel, timing-only ablations by bit mask (C3_ABM: 1 no LDS-DMA, 2 no MFMA, 4 no barrier, 8 no fragment reads, 16 same tile,
# 32 contiguous patch addresses, 64 no vmcnt wait): single layers, 8 views bf16
V=nerffaceediting_amd/csrc/build/variants
for name in ${VARIANTS:-shipped abm64 abm32 abm68 abm8}; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  echo "== $name"; NFE_RENDER_LIB=$lib python3 tools/time_conv.py bf16 8 2>&1 | grep -v "^/opt" | head -${LINES_:-2}
done
