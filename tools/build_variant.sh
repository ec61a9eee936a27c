#!/bin/bash
# Experiment build of libnfe_render.so with extra compiler flags, through the regular pipeline (csrc/Makefile, incl. pk_opsel_fix.py):
#   tools/build_variant.sh <name> [<file, ignored: kept for old call sites>] [flags...]
#   ->  nerffaceediting_amd/csrc/build/variants/<name>.so   (use with NFE_RENDER_LIB)
set -e
name=$1; shift
case "$1" in nfe_render|nfe_render_bwd|nfe_dense|nfe_planes) shift;; esac
cd "$(dirname "$0")/../nerffaceediting_amd/csrc"
mkdir -p build/variants
make -s -j4 BUILD=build/variants/obj_$name OUT=build/variants/$name.so EXTRA="$*" > build/variants/$name.log 2>&1 || { tail -30 build/variants/$name.log; exit 1; }
grep pk_opsel_fix build/variants/$name.log | tr "\n" ";"; rm -f build/variants/$name.log
rm -rf build/variants/obj_$name
echo built build/variants/$name.so
