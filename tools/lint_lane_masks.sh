#!/bin/bash
# Counts, per kernel, the instruction pattern behind profiles/experiments/r02_lane_mask.md: a lane mask combined on the scalar unit
# (s_and/or/andn2/xor/xnor_b64 vcc, s[..]) and consumed by a v_cndmask_b32_e32 within three instructions.  The render and backward
# kernels are kept at zero.   tools/lint_lane_masks.sh [file.hip ...]
cd "$(dirname "$0")/.."
files=${@:-nerffaceediting_amd/csrc/nfe_render.hip nerffaceediting_amd/csrc/nfe_render_bwd.hip}
rc=0
for f in $files; do
  s=/tmp/lint_$(basename $f).s
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Iinclude -Inerffaceediting_amd/csrc -x hip -c $f --cuda-device-only -S -o $s || exit 2
  n=$(awk '/^_ZN3nfe/{k=$1} /s_(and|or|andn2|xor|xnor)_b64 vcc, s\[/{pend=3; next} pend>0 && /v_cndmask_b32_e32.*vcc/{c[k]++} {if(pend>0)pend--} END{t=0; for(k in c){print c[k], k; t+=c[k]} print "TOTAL", t}' $s | tee /dev/stderr | awk '/^TOTAL/{print $2}')
  echo "$f: $n"
  [ "$n" = "0" ] || rc=1
done
exit $rc
