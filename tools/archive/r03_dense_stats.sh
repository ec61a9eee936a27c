#!/bin/bash
# Per-kernel time of the full path (config 3: 8 views, bf16 convs; and the FFHQ configuration: 4 views, split-bf16)
export TMPDIR=/tmp
OUT=gpurun_out/r03_dense_stats
mkdir -p $OUT
for w in full ffhq; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --preroll-s 0.2 --streams 1 > $OUT/bench_$w.log 2>&1
  find $OUT/st_$w -name "*kernel_stats.csv" -exec cp {} $OUT/ks_$w.csv \;
  rm -rf $OUT/st_$w
  tail -1 $OUT/bench_$w.log | cut -c1-300
done
