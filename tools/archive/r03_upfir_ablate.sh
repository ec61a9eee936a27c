#!/bin/bash
# upfir_kernel: timing-only ablations (no scratch reads / no stores) on config 3 (8 views, bf16), single stream
export TMPDIR=/tmp
OUT=gpurun_out/r03_upfir
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
for v in ${VARIANTS:-default upfir_abl1 upfir_abl2}; do
  lib=$V/$v.so; [ $v = default ] && lib=nerffaceediting_amd/libnfe_render.so
  NFE_RENDER_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$v -- python3 bench.py --workload ${WL:-full} --steps 10 --warmup 3 --preroll-s 0.2 --streams 1 > $OUT/bench_$v.log 2>&1
  find $OUT/st_$v -name "*kernel_stats.csv" -exec cp {} $OUT/ks_$v.csv \;
  rm -rf $OUT/st_$v
  echo "== $v: $(grep -h 'upfir_kernel\|conv3_kernel<1, 1, true' $OUT/ks_$v.csv | cut -d, -f1-4 | tr '\n' ' ')"
done
