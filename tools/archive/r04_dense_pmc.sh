#!/bin/bash
# Round 4: PMC passes + kernel stats of the dense path in fp16 operand arithmetic (FFHQ configuration, 4 views) and in bf16 (config 3
# shape, 8 views): the conv3_kernel<2, ...> instantiations are new this round, the bf16 ones are re-measured beside them.
export TMPDIR=/tmp
OUT=gpurun_out/r04_dense
mkdir -p $OUT
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<2, 2, false, 2, 8, 2" bash tools/pmc.sh r04_dense/pmc_fp16 4 128 48 48 fp16 > $OUT/r04_pmc_dense_fp16.txt 2>&1
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<1, 2, false, 2, 8, 2" bash tools/pmc.sh r04_dense/pmc_bf16 8 128 64 0 bf16 > $OUT/r04_pmc_dense_bf16.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload ffhq --conv-math fp16 --steps 10 --warmup 2 > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats_dense_ffhq_fp16.csv \;
rm -rf $OUT/stats $OUT/pmc_fp16/*/ $OUT/pmc_bf16/*/
head -14 $OUT/r04_kernel_stats_dense_ffhq_fp16.csv | cut -c1-150
python3 - <<'PY'
import re
for f in ("gpurun_out/r04_dense/r04_pmc_dense_fp16.txt", "gpurun_out/r04_dense/r04_pmc_dense_bf16.txt"):
    cur = None; rows = {}
    for ln in open(f):
        m = re.match(r"== (.*?)\s+dispatches", ln)
        if m: cur = m.group(1)[:70]; rows[cur] = {}
        m = re.match(r"\s+(\w+)\s+avg/dispatch ([\d.e+]+)", ln)
        if m and cur: rows[cur][m.group(1)] = float(m.group(2))
    print(f)
    for k, v in rows.items():
        if "conv3_kernel" in k and "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            print("  %-72s mfma_busy %.2f  lds_conflict %.3g  coexec %.2f" % (k, v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(v["SQ_VALU_MFMA_BUSY_CYCLES"], 1)))
PY
