#!/bin/bash
# Round 4, final build, ONE box: every bench line + the kernel-trace statistics of the default bench command (without the strong-scaling
# job, whose launches have other sizes), of the two-pass and of the FFHQ / config-3 workloads on one stream
# (-> profiles/r04_bench_line*.json, profiles/r04_kernel_stats*.csv)
export TMPDIR=/tmp
bash tools/r04_lines.sh > gpurun_out/r04_lines_summary.txt 2>&1
OUT=gpurun_out/r04_final; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling > $OUT/bench_render.log 2>&1
find $OUT/s1 -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats.csv \; ; rm -rf $OUT/s1
for w in twopass ffhq full; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s2 -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $OUT/bench_$w.log 2>&1
  find $OUT/s2 -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats_$w.csv \; ; rm -rf $OUT/s2
done
cat gpurun_out/r04_lines_summary.txt
head -3 $OUT/r04_kernel_stats.csv | cut -c1-200
