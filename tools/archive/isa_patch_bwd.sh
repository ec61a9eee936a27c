#!/bin/bash
# (works on hipcc's RAW assembly on purpose - pk_opsel_fix.py is NOT applied - so that the hazard it removes can be studied)
# ISA-level experiments on the CURRENT nfe_render_bwd.hip (round 4: profiles/experiments/r04_bwd_lanes48_rootcause.md): compile to AMDGPU
# assembly, run a Python patcher over it, assemble + link it back into a variant library beside the regular objects.
#   tools/isa_patch_bwd.sh <name> <patch.py|none> [patcher args / hipcc flags after --]   -> nerffaceediting_amd/csrc/build/variants/<name>.so
set -e
name=$1; patch=$2; shift 2
pargs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do pargs+=("$1"); shift; done; [ "$1" = "--" ] && shift
F=${ISA_FILE:-nfe_render_bwd}          # ISA_FILE=nfe_render: the same round trip for another file of the library
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/nerffaceediting_amd/csrc
W=$C/build/variants/isa_$name
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p $W
make -s -C $C -j4 > /dev/null
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I$ROOT/include -I$C $*"
/opt/rocm/bin/hipcc $FL -x hip $C/$F.hip --cuda-device-only -S -o $W/dev.s 2>/dev/null
if [ "$patch" = none ]; then cp $W/dev.s $W/dev_p.s; else python3 $patch $W/dev.s $W/dev_p.s "${pargs[@]}"; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_p.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FL -x hip $C/$F.hip --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c -o $W/$F.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/variants/$name.so $C/build/nfe_api.cpp.o $(for f in nfe_render nfe_render_bwd nfe_planes nfe_dense; do if [ $f = $F ]; then echo $W/$F.hip.o; else echo $C/build/$f.hip.o; fi; done)
rm -rf $W
echo built $C/build/variants/$name.so
