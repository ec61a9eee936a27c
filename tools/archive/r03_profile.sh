#!/bin/bash
# Round 3, one box, one build: PMC passes of the headline kernel -> profiles/r03_issue_floor.json -> bench lines of every workload
# -> rocprofv3 kernel stats of the headline command -> PMC passes of the two dense workloads.
#   gpurun -- 'bash tools/r03_profile.sh'   then copy gpurun_out/r03_profile/r03_* into profiles/
export TMPDIR=/tmp
OUT=gpurun_out/r03_profile
mkdir -p $OUT
PMC_KERNEL="render_kernel<false, false, 0, false, false, false, true, false, false>" bash tools/pmc.sh r03_profile/pmc > $OUT/r03_pmc_render.txt 2>&1
cp $OUT/pmc/issue_floor.json profiles/r03_issue_floor.json
cp $OUT/pmc/issue_floor.json $OUT/r03_issue_floor.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_line.json 2> $OUT/bench.err
for w in full ffhq twopass editstep orbit; do
  python3 bench.py --workload $w --steps 10 --warmup 2 > $OUT/r03_bench_line_$w.json 2>> $OUT/bench.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r03_kernel_stats.csv \;
rm -rf $OUT/stats
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<1, 2, false, 2, 8, 2" bash tools/pmc.sh r03_profile/pmc_bf16 8 128 64 0 bf16 > $OUT/r03_pmc_dense_bf16.txt 2>&1
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<3, 2, false, 1, 4, 4" bash tools/pmc.sh r03_profile/pmc_x3 4 128 48 48 bf16x3 > $OUT/r03_pmc_dense_x3.txt 2>&1
rm -rf $OUT/pmc/*/ $OUT/pmc_bf16/*/ $OUT/pmc_x3/*/
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_profile/r03_bench_line*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["unit"], "ms/step", round(d["ms_per_step"], 3), d.get("strong_scaling", {}).get("views_per_s"), d["config"].get("stage_ms"), d["roofline"].get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -5 $OUT/r03_kernel_stats.csv | cut -c1-220
