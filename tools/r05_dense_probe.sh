#!/bin/bash
# round 5, dense path: per-layer kernel times of config 3 (8 views, bf16) and A/B of library variants on tools/time_full.py
#   tools/r05_dense_probe.sh [variant.so ...]
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_dense
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_dense/trace -- python3 tools/time_full.py 8 512 64 0 bf16 > gpurun_out/r05_dense/trace.log 2>&1
python3 tools/layer_times.py gpurun_out/r05_dense/trace > gpurun_out/r05_dense/layers.txt 2>&1
tail -70 gpurun_out/r05_dense/layers.txt
for rep in 1 2; do
  for lib in nerffaceediting_amd/libnfe_render.so "$@"; do
    echo "== $lib"
    NFE_RENDER_LIB=$PWD/$lib python3 tools/time_full.py 8 512 64 0 bf16 2>&1 | tail -3
    NFE_RENDER_LIB=$PWD/$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | tail -2
  done
done 2>&1 | tee gpurun_out/r05_dense/ab.txt
