#!/bin/bash
# Round 4, final build: PMC passes + kernel stats of the headline command and of the two-pass workload (all render kernels are the
# wave-specialised ones now), then every bench line, on ONE box.
export TMPDIR=/tmp
OUT=gpurun_out/r04_p3
mkdir -p $OUT
bash tools/pmc.sh r04_p3/pmc > $OUT/pmc_default.txt 2>&1
PMC_KERNEL="render_ws_kernel<4, 2, true, false, false, false>" python3 tools/pmc_summary.py $OUT/pmc > $OUT/r04_pmc_render_ws.txt 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_issue_floor.json
PMC_KERNEL="render_kernel<false, false, 1," python3 tools/pmc_summary.py $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_issue_floor_fp32.json
bash tools/pmc.sh r04_p3/pmc2 --workload twopass --steps 3 --warmup 1 > $OUT/pmc_twopass.txt 2>&1
PMC_KERNEL="render_ws_kernel<4, 2, true, true, true, false>" python3 tools/pmc_summary.py $OUT/pmc2 > $OUT/r04_pmc_twopass.txt 2>&1
cp $OUT/pmc2/issue_floor.json $OUT/r04_issue_floor_twopass_final.json
PMC_KERNEL="render_ws_kernel<4, 2, true, false, false, true>" python3 tools/pmc_summary.py $OUT/pmc2 > /dev/null 2>&1
cp $OUT/pmc2/issue_floor.json $OUT/r04_issue_floor_twopass_sigma.json
PMC_KERNEL="importance_kernel" python3 tools/pmc_summary.py $OUT/pmc2 > /dev/null 2>&1
cp $OUT/pmc2/issue_floor.json $OUT/r04_issue_floor_twopass_importance.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 bench.py --workload twopass --steps 10 --warmup 2 > $OUT/stats2.log 2>&1
find $OUT/stats2 -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats_twopass.csv \;
rm -rf $OUT/stats $OUT/stats2 $OUT/pmc/*/ $OUT/pmc2/*/
head -4 $OUT/r04_kernel_stats.csv | cut -c1-180; head -5 $OUT/r04_kernel_stats_twopass.csv | cut -c1-180
grep -h '"kernel"\|avg_ns_profiled' $OUT/r04_issue_floor*.json
