#!/bin/bash
# Binned scatter (default) vs the sorted-atomic form: parity suite, timing, ablations of the decoder-backward kernel, kernel stats.
export TMPDIR=/tmp
OUT=gpurun_out/r02_bwdbin
mkdir -p $OUT
python3 -m pytest tests/test_render_backward_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/tests.txt
cat $OUT/tests.txt
rm -f $OUT/time.txt
for mode in sorted binned; do
  echo "== NFE_BWD_SCATTER=$mode" >> $OUT/time.txt
  NFE_BWD_SCATTER=$mode python3 tools/time_backward.py 4 128 48 48 256 2>&1 | grep -v "^/opt" | tail -2 >> $OUT/time.txt
done
V=nerffaceediting_amd/csrc/build/variants
for v in bwd_abl5 bwd_abl6; do
  [ -f $V/$v.so ] || continue
  echo "== variant $v (5: all taps read one texel row; 6: no df store / records)" >> $OUT/time.txt
  NFE_RENDER_LIB=$V/$v.so python3 tools/time_backward.py 4 128 48 48 256 2>&1 | grep -v "^/opt" | tail -1 >> $OUT/time.txt
done
cat $OUT/time.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
head -9 $OUT/kernel_stats.csv | cut -c1-160
python3 tools/fuzz_sweep.py 900 0 120 2>&1 | grep -v "^/opt" | tail -3 | tee $OUT/fuzz_bwd.txt
