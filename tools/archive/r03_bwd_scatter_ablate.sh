#!/bin/bash
# decoder-backward kernel with both plane sets as leaves: timing-only ablations (5: every tap reads one texel row; 6: no df rows / records)
export TMPDIR=/tmp
OUT=gpurun_out/r03_bwd_scatter
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
for v in shipped bwd_abl5 bwd_abl6; do
  lib=$V/$v.so; [ $v = shipped ] && lib=nerffaceediting_amd/libnfe_render.so
  BOTH_ONLY=1 NFE_RENDER_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/$v.log 2>&1
  find $OUT/st -name "*kernel_stats.csv" -exec cp {} $OUT/ks_$v.csv \;
  rm -rf $OUT/st
  echo "== $v: $(grep -h 'bwd_scatter_sorted\|accumulate' $OUT/ks_$v.csv | cut -d, -f1,4 | tr '\n' ' ')"
done
