#!/bin/bash
# Register-tile accumulate pass: index-mode window per 4 records (default) vs per record, and timing-only ablations.
export TMPDIR=/tmp
OUT=gpurun_out/r03_bwdacc2
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
python3 -m pytest tests/test_render_backward_gpu.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.txt; cat $OUT/tests.txt
for v in ${VARIANTS:-default acc_g1 acc_abl1 acc_abl2}; do
  lib=$V/$v.so; [ $v = default ] && lib=nerffaceediting_amd/libnfe_render.so
  NFE_RENDER_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$v -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/st_$v.log 2>&1
  find $OUT/st_$v -name "*kernel_stats.csv" -exec cp {} $OUT/ks_$v.csv \;
  rm -rf $OUT/st_$v
  echo "== $v: $(grep accumulate $OUT/ks_$v.csv | cut -d, -f1,4)"
done
