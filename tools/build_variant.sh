#!/bin/bash
# Variant of libnfe_render.so that differs in ONE .hip file (the other objects come from the regular build):
#   tools/build_variant.sh <name> <nfe_render|nfe_render_bwd|nfe_dense|nfe_planes> [flags...]
#   ->  nerffaceediting_amd/csrc/build/variants/<name>.so   (use with NFE_RENDER_LIB)
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../nerffaceediting_amd/csrc"
make -s -j4 > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -I. "$@" -x hip -c $file.hip -o build/variants/$name.$file.o
objs=""
for f in nfe_render nfe_render_bwd nfe_planes nfe_dense; do
  if [ $f = $file ]; then objs="$objs build/variants/$name.$file.o"; else objs="$objs build/$f.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so build/nfe_api.cpp.o $objs
echo built build/variants/$name.so
