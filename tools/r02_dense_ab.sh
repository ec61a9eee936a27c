#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_dense
mkdir -p $OUT
python3 -m pytest tests/test_dense_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tail -4 > $OUT/tests_ks.txt
cat $OUT/tests_ks.txt
for k in 0 1 0 1; do
  echo "== NFE_C3_KSPLIT=$k" >> $OUT/ab_ks.txt
  NFE_C3_KSPLIT=$k python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" >> $OUT/ab_ks.txt
  NFE_C3_KSPLIT=$k python3 tools/time_full.py 1 128 48 48 bf16x3 2>&1 | grep -E "^N=" >> $OUT/ab_ks.txt
done
cat $OUT/ab_ks.txt
