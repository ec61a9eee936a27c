#!/bin/bash
# Round 4, profile of the shipped build: PMC passes of the headline command (render_ws_kernel is the dominant kernel now), the
# fused kernel beside it (NFE_RENDER_WS=0), the list of SQ instruction-class counters this rocprofv3 knows, bench lines.
export TMPDIR=/tmp
OUT=gpurun_out/r04_p2
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_INSTS_VALU[A-Z0-9_]*|SQ_INST_CYCLES[A-Z0-9_]*|SQ_VALU[A-Z0-9_]*|SQ_ACTIVE_INST[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt
wc -l $OUT/sq_counters.txt
bash tools/pmc.sh r04_p2/pmc > $OUT/pmc_default.txt 2>&1
PMC_KERNEL="render_ws_kernel<4, 2, true, false>" python3 tools/pmc_summary.py $OUT/pmc > $OUT/r04_pmc_render_ws.txt 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_issue_floor.json
NFE_RENDER_WS=0 bash tools/pmc.sh r04_p2/pmc0 > $OUT/pmc_fused.txt 2>&1
PMC_KERNEL="render_kernel<false, false, 0, false, false, false, true, false, false>" python3 tools/pmc_summary.py $OUT/pmc0 > $OUT/r04_pmc_render_fused.txt 2>&1
cp $OUT/pmc0/issue_floor.json $OUT/r04_issue_floor_fused.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_line.json 2> $OUT/bench.err
NFE_RENDER_WS=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling > $OUT/r04_bench_line_fused.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats.csv \;
rm -rf $OUT/stats $OUT/pmc/*/ $OUT/pmc0/*/
head -4 $OUT/r04_kernel_stats.csv | cut -c1-200
for f in $OUT/r04_bench_line.json $OUT/r04_bench_line_fused.json; do python3 - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["value"], "ms/step", round(d["ms_per_step"], 3), "kernel_ms", d["roofline"].get("kernel_ms"), d["roofline"].get("frac"))
PY
done
