#!/bin/bash
# round 6: per-layer kernel times of synthesis() at config 3's shape (8 views bf16) and at the FFHQ shape (4 views split-bf16):
# rocprofv3 kernel trace of tools/time_full.py, the kernels of the LAST synthesis call in launch order (tools/layer_times.py)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_layers
for cfg in "8 512 64 0 bf16" "4 128 48 48 bf16x3"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf gpurun_out/r06_layers/tr
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_layers/tr -- python3 tools/time_full.py $cfg > /dev/null 2>&1
  python3 tools/layer_times.py gpurun_out/r06_layers/tr > gpurun_out/r06_layers/layer_times_$tag.txt
  rm -rf gpurun_out/r06_layers/tr
  echo "== $cfg"; cat gpurun_out/r06_layers/layer_times_$tag.txt
done
