#!/bin/bash
# conv3_kernel: is the operand staging bound by where the operands come from?  Ablations 5 / 6 make every workgroup stage the
# same tile (L2-resident operands), with and without the MFMAs; 2 = no MFMA with the real addresses.  SR conv 256^2 256->256, 8 views.
V=nerffaceediting_amd/csrc/build/variants
mkdir -p gpurun_out/r03_ablate
for name in shipped abl2 abl5 abl6; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-170
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|$name |" | cut -c1-170
done 2>&1 | tee gpurun_out/r03_ablate/ablate2.log
