#!/bin/bash
# Build timing-only variants of the render kernel (outputs are wrong by construction) and time them
# against the real build in one process each.  Usage (on the GPU box): tools/ablate.sh
set -e
cd nerffaceediting_amd/csrc
for v in GATHER MLP; do
  mkdir -p build_$v
  for f in nfe_api.cpp nfe_render.hip nfe_planes.hip nfe_dense.hip; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DNFE_ABLATE_$v -x hip -c $f -o build_$v/$f.o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libnfe_ablate_$v.so build_$v/*.o
done
cd ../..
echo "full:";        python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | grep -oE '"kernel_ms": [0-9.]+'
echo "no gather:";   NFE_RENDER_LIB=/tmp/libnfe_ablate_GATHER.so python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | grep -oE '"kernel_ms": [0-9.]+'
echo "no decoder:";  NFE_RENDER_LIB=/tmp/libnfe_ablate_MLP.so python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline | grep -oE '"kernel_ms": [0-9.]+'
