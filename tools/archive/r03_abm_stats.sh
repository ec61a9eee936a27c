#!/bin/bash
# per-kernel times of config 3 / FFHQ with an ablation library (single stream):  VARIANTS="shipped abm128" WL=full bash tools/r03_abm_stats.sh
export TMPDIR=/tmp
OUT=gpurun_out/r03_abm_stats
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
for v in ${VARIANTS:-shipped abm128}; do
  lib=$V/$v.so; [ $v = shipped ] && lib=nerffaceediting_amd/libnfe_render.so
  for w in ${WL:-full ffhq}; do
    NFE_RENDER_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 bench.py --workload $w --steps 10 --warmup 3 --preroll-s 0.2 --streams 1 > $OUT/bench_${v}_$w.log 2>&1
    find $OUT/st -name "*kernel_stats.csv" -exec cp {} $OUT/ks_${v}_$w.csv \;
    rm -rf $OUT/st
    echo "== $v $w"; grep -h 'conv3_kernel\|upfir\|torgb\|rgb_combine' $OUT/ks_${v}_$w.csv | cut -d, -f1-4 | cut -c1-120
  done
done
