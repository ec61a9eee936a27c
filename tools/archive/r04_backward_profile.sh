#!/bin/bash
# Round 4: the edit step's backward, ONE mode (both plane sets are leaves), on ONE box: kernel trace (durations) and PMC passes (traffic)
# of the same command, so that bwd_accumulate_reg_kernel and bwd_scatter_sorted_kernel get ONE duration each.
export TMPDIR=/tmp
OUT=gpurun_out/r04_bwd
mkdir -p $OUT
BOTH_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats_backward.csv \;
BOTH_ONLY=1 PMC_PROG=tools/time_backward.py PMC_KERNEL="bwd_accumulate_reg" bash tools/pmc.sh r04_bwd/pmc 4 128 48 48 256 > $OUT/r04_pmc_backward.txt 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_backward_accumulate_counters.json
BOTH_ONLY=1 PMC_KERNEL="bwd_scatter_sorted_kernel<true, true>" python3 tools/pmc_summary.py $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/issue_floor.json $OUT/r04_backward_scatter_counters.json
rm -rf $OUT/stats $OUT/pmc/*/
head -9 $OUT/r04_kernel_stats_backward.csv | cut -c1-160
python3 - <<'PY'
import json
for n in ("accumulate", "scatter"):
    d = json.load(open(f"gpurun_out/r04_bwd/r04_backward_{n}_counters.json"))
    hbm = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
    print(n, d["kernel"][:60], "avg_ns under PMC", round(d["avg_ns_profiled"]), "HBM GB", round(hbm / 1e9, 2), "TB/s", round(hbm / d["avg_ns_profiled"] / 1e3, 2), "dispatches", d["dispatches"])
PY
