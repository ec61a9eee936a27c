#!/bin/bash
# One box, one build: PMC passes -> profiles/r02_issue_floor.json -> un-profiled bench line -> rocprofv3 kernel stats.
#   gpurun -- 'bash tools/r02_profile.sh'   then copy gpurun_out/r02_profile/* into profiles/
export TMPDIR=/tmp
OUT=gpurun_out/r02_profile
mkdir -p $OUT
PMC_KERNEL="render_kernel<false, false, 0, false, false, false, true, false>" bash tools/pmc.sh r02_profile/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/issue_floor.json profiles/r02_issue_floor.json
cp $OUT/pmc/issue_floor.json $OUT/r02_issue_floor.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r02_bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r02_kernel_stats.csv \;
rm -rf $OUT/stats/*/*kernel_trace.csv
head -c 1500 $OUT/r02_bench_line.json; echo; head -4 $OUT/r02_kernel_stats.csv | cut -c1-200
