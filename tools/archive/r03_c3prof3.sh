#!/bin/bash
# epilogue cycles per wave (light profile) of the shipped plain kernels and of timing-only ablations: 256 no constant reads, 512 no stores
V=nerffaceediting_amd/csrc/build/variants
for v in c3prof2 c3p_256 c3p_512 c3p_768; do
  for m in bf16 bf16x3; do echo "== $v $m"; NFE_RENDER_LIB=$V/$v.so python3 tools/c3_profile.py $m 8 2>&1 | grep -v "^/opt" | head -2 | cut -c1-30,95-150; done
done
