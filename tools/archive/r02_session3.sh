#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_s3
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
for n in square_rt10; do
  for b in 2; do
  echo "== variant $n blocks_per_cu=$b" >> $OUT/hash10.txt
  HASH_SQUARE_ONLY=1 NFE_RENDER_LIB=$V/$n.so NFE_RENDER_BLOCKS_PER_CU=$b python3 tools/hash_occupancy.py 6 2>&1 | grep -E "CASE" >> $OUT/hash10.txt
  done
done
cat $OUT/hash10.txt
