#!/bin/bash
V=nerffaceediting_amd/csrc/build/variants
mkdir -p gpurun_out/r03_ab2
run() { # name lib env
  NFE_RENDER_LIB=$2 env $3 python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|$1 |" | cut -c1-175
  NFE_RENDER_LIB=$2 env $3 python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|$1 |" | cut -c1-175
}
{
run shipped "" "X=1"
run wide8 "" "NFE_C3_WIDE8=1"
run s3x2 $V/s3x2.so "X=1"
run s3x2+wide8 $V/s3x2.so "NFE_C3_WIDE8=1"
run s2x2 $V/s2x2.so "X=1"
run shipped "" "X=1"
for v in s3x2 s2x2; do NFE_RENDER_LIB=$V/$v.so python3 -m pytest tests/test_dense_gpu.py -m gpu -x -q 2>&1 | tail -1 | sed "s|^|$v parity: |"; done
} 2>&1 | tee gpurun_out/r03_ab2/ab.log
