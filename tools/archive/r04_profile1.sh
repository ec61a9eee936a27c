#!/bin/bash
# Round 4, GPU call 1: new GPU tests (one-rank RCCL, sample-colour layout), the default bench line (this round's baseline), PMC
# passes of the headline workload (split-bf16 AND exact-fp32 render kernels: both run in the default command) and of the two-pass
# workload (sigma pass, importance kernel, final dual-set pass) -> profiles/r04_issue_floor*.json
export TMPDIR=/tmp
OUT=gpurun_out/r04_p1
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_rccl_single_gpu.py tests/test_render_backward_gpu.py -x -q -m gpu -k "rccl or collective or sample_colors or kept" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_line_base.json 2> $OUT/bench.err
python3 bench.py --workload twopass --steps 10 --warmup 2 > $OUT/r04_bench_line_twopass_base.json 2>> $OUT/bench.err
bash tools/pmc.sh r04_p1/pmc > $OUT/pmc_default.txt 2>&1
PMC_KERNEL="render_kernel<false, false, 0, false, false, false, true, false, false>" python3 tools/pmc_summary.py $OUT/pmc > $OUT/r04_pmc_render.txt 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_issue_floor.json
PMC_KERNEL="render_kernel<false, false, 1," python3 tools/pmc_summary.py $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_issue_floor_fp32.json
bash tools/pmc.sh r04_p1/pmc2 --workload twopass --steps 3 --warmup 1 > $OUT/pmc_twopass.txt 2>&1
PMC_KERNEL="render_kernel<true, false, 0," python3 tools/pmc_summary.py $OUT/pmc2 > $OUT/r04_pmc_twopass.txt 2>&1
cp $OUT/pmc2/issue_floor.json $OUT/r04_issue_floor_twopass_final.json
PMC_KERNEL="render_kernel<true, true, 0," python3 tools/pmc_summary.py $OUT/pmc2 > /dev/null 2>&1
cp $OUT/pmc2/issue_floor.json $OUT/r04_issue_floor_twopass_sigma.json
PMC_KERNEL="importance_kernel" python3 tools/pmc_summary.py $OUT/pmc2 > /dev/null 2>&1
cp $OUT/pmc2/issue_floor.json $OUT/r04_issue_floor_twopass_importance.json
rm -rf $OUT/pmc/*/ $OUT/pmc2/*/
for f in $OUT/r04_bench_line*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["value"], d["unit"], "ms/step", round(d["ms_per_step"], 3), d["roofline"].get("kernel_ms"), d["roofline"].get("frac"))
PY
done
grep -h '"kernel"\|avg_ns_profiled' $OUT/r04_issue_floor*.json
