#!/bin/bash
# Round 4: dynamic instruction-class counters (SQ_INSTS_VALU_TRANS_F32 ..., SQ_VALU_MFMA_COEXEC_CYCLES) of the final build's render
# kernels, wave-specialised and fused, to check the static census and the no-overlap model against the hardware's own counts.
export TMPDIR=/tmp
OUT=gpurun_out/r04_p4
mkdir -p $OUT
bash tools/pmc.sh r04_p4/pmc > $OUT/pmc_default.txt 2>&1
PMC_KERNEL="render_ws_kernel<4, 2, true, false, false, false>" python3 tools/pmc_summary.py $OUT/pmc > $OUT/r04_pmc_render_ws.txt 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_issue_floor.json
NFE_RENDER_WS=0 bash tools/pmc.sh r04_p4/pmc0 > $OUT/pmc_fused.txt 2>&1
PMC_KERNEL="render_kernel<false, false, 0, false, false, false, true, false, false>" python3 tools/pmc_summary.py $OUT/pmc0 > $OUT/r04_pmc_render_fused.txt 2>&1
cp $OUT/pmc0/issue_floor.json $OUT/r04_issue_floor_fused.json
rm -rf $OUT/pmc/*/ $OUT/pmc0/*/
python3 - <<'PY'
import json
for n in ("", "_fused"):
    d = json.load(open(f"gpurun_out/r04_p4/r04_issue_floor{n}.json"))
    steps = 32768 * 64
    print(n or "ws", d["kernel"][:50], {k: round(d[k] / steps, 1) for k in sorted(d) if k.startswith(("SQ_INSTS", "SQ_VALU_MFMA", "SQ_INST_CYCLES"))})
PY
