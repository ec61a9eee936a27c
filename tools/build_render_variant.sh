#!/bin/bash
# Variant of libnfe_render.so that differs in nfe_render.hip only (the other objects come from the regular build):
#   tools/build_render_variant.sh <name> [flags...]  ->  nerffaceediting_amd/csrc/build/variants/<name>.so  (use with NFE_RENDER_LIB)
set -e
name=$1; shift
cd "$(dirname "$0")/../nerffaceediting_amd/csrc"
make -s -j4 > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -I. "$@" -x hip -c nfe_render.hip -o build/variants/$name.render.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so build/nfe_api.cpp.o build/variants/$name.render.o build/nfe_render_bwd.hip.o build/nfe_planes.hip.o build/nfe_dense.hip.o
echo built build/variants/$name.so
