#!/bin/bash
# Build a variant of libnfe_render.so with extra compiler flags: tools/build_variant.sh <name> [flags...]
# -> nerffaceediting_amd/csrc/build/variants/<name>.so (use with NFE_RENDER_LIB)
set -e
name=$1; shift
cd "$(dirname "$0")/../nerffaceediting_amd/csrc"
mkdir -p build/variants/obj_$name
for f in nfe_api.cpp nfe_render.hip nfe_render_bwd.hip nfe_planes.hip nfe_dense.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -I. "$@" -x hip -c $f -o build/variants/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so build/variants/obj_$name/*.o
echo built build/variants/$name.so
