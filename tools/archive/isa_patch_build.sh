#!/bin/bash
# ISA-level delta debugging of nfe_render_bwd.hip (lane-mask finding): compile the file to AMDGPU assembly with the given flags,
# run a Python patcher over the assembly, assemble + link it back into a variant library next to the other (unpatched) objects.
#   tools/isa_patch_build.sh <name> <patch.py|none> [hipcc flags...]     -> nerffaceediting_amd/csrc/build/variants/<name>.so
# The patcher gets (in.s, out.s) as arguments.  Needs build/variants/obj_tapsc/ (tools/build_variant.sh tapsc -DNFE_TAPS_COMBINED=1).
set -e
name=$1; patch=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/nerffaceediting_amd/csrc
W=$C/build/variants/isa_$name
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p $W
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I$ROOT/include -I$C $*"
/opt/rocm/bin/hipcc $FL -x hip $C/nfe_render_bwd.hip --cuda-device-only -S -o $W/dev.s 2>/dev/null
if [ "$patch" = none ]; then cp $W/dev.s $W/dev_p.s; else python3 $patch $W/dev.s $W/dev_p.s; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_p.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FL -x hip $C/nfe_render_bwd.hip --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c -o $W/nfe_render_bwd.hip.o 2>/dev/null
objs=$(ls $C/build/variants/obj_tapsc/*.o | grep -v nfe_render_bwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/variants/$name.so $objs $W/nfe_render_bwd.hip.o
echo built $C/build/variants/$name.so
