#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_bwd
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
rm -f $OUT/abl.txt
for lib in "" $V/bwd_abl3.so $V/bwd_abl4.so; do
for d in valu mfma; do
  echo "== lib=${lib:-shipped} NFE_BWD_DECODER=$d" >> $OUT/abl.txt
  NFE_RENDER_LIB=$lib NFE_BWD_DECODER=$d python3 tools/time_backward.py 1 128 48 48 256 2>&1 | grep -v "^/opt" | tail -1 >> $OUT/abl.txt
done; done
cat $OUT/abl.txt
