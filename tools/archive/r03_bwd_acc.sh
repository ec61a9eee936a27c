#!/bin/bash
# Accumulate pass of the binned backward: tile in registers (default) vs in LDS (NFE_BWD_ACC=lds): parity suite, timing, kernel stats.
export TMPDIR=/tmp
OUT=gpurun_out/r03_bwdacc
mkdir -p $OUT
python3 -m pytest tests/test_render_backward_gpu.py tests/test_e2e_gpu.py -m gpu -q -k "backward or grad or differentiable or edit" 2>&1 | tail -5 > $OUT/tests.txt
cat $OUT/tests.txt
rm -f $OUT/time.txt
for mode in lds reg; do
  echo "== NFE_BWD_ACC=$mode" >> $OUT/time.txt
  NFE_BWD_ACC=$mode python3 tools/time_backward.py 4 128 48 48 256 2>&1 | grep -v "^/opt" | tail -2 >> $OUT/time.txt
  NFE_BWD_ACC=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$mode -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/stats_$mode.log 2>&1
  find $OUT/stats_$mode -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$mode.csv \;
  rm -rf $OUT/stats_$mode
  head -9 $OUT/kernel_stats_$mode.csv | cut -c1-150
done
cat $OUT/time.txt
