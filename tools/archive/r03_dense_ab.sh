#!/bin/bash
# round 3 dense A/B: variant libraries (tools/build_variant.sh) against the shipped one: parity of the dense tests, stage times of
# BASELINE config 3 (8 views, bf16) and the FFHQ configuration (4 views, bf16x3).   tools/r03_dense_ab.sh name1 name2 ...
export TMPDIR=/tmp
OUT=gpurun_out/r03_dense_ab
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
for name in shipped "$@" shipped; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  if [ "$name" != shipped ]; then
    NFE_RENDER_LIB=$lib python3 -m pytest tests/test_dense_gpu.py -m gpu -x -q 2>&1 | tail -1 | sed "s|^|$name parity: |"
  fi
  for rep in 1 2; do
    NFE_RENDER_LIB=$lib python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|$name |"
  done
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|$name |"
done 2>&1 | tee $OUT/ab_$(date +%H%M%S).log
