#!/bin/bash
# Round 5: the edit step's backward (4 views of 128^2 x (48+48), both plane sets as leaves) on ONE box: kernel trace (durations) and PMC
# passes (counters) of the same command, for the wave-specialised decoder-backward kernel (default) and, durations only, for round 4's
# single-wave kernel (NFE_BWD_DECODER=single).  Writes gpurun_out/r06_bwd/{r06_kernel_stats_backward.csv, r06_kernel_stats_backward_single.csv,
# r06_pmc_backward.txt, r06_backward_counters.json}; also times the SR-head gradient (tools/time_sr_grad.py 4).
export TMPDIR=/tmp
OUT=gpurun_out/r06_bwd
mkdir -p $OUT
BOTH_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r06_kernel_stats_backward.csv \;
rm -rf $OUT/stats
NFE_BWD_DECODER=single BOTH_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/time_backward.py 4 128 48 48 256 > $OUT/stats_single.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r06_kernel_stats_backward_single.csv \;
rm -rf $OUT/stats
BOTH_ONLY=1 PMC_PROG=tools/time_backward.py PMC_KERNEL="bwd_decoder_kernel" bash tools/pmc.sh r06_bwd/pmc 4 128 48 48 256 > $OUT/r06_pmc_backward.txt 2>&1
cp $OUT/pmc/issue_floor.json $OUT/decoder.json
BOTH_ONLY=1 PMC_KERNEL="bwd_accumulate_reg" python3 tools/pmc_summary.py $OUT/pmc > /dev/null 2>&1
cp $OUT/pmc/issue_floor.json $OUT/accumulate.json
rm -rf $OUT/pmc/*/
python3 tools/time_backward.py 4 128 48 48 256 > $OUT/time_backward.txt 2>&1
NFE_BWD_DECODER=single python3 tools/time_backward.py 4 128 48 48 256 >> $OUT/time_backward.txt 2>&1
python3 tools/time_sr_grad.py 4 > $OUT/time_sr_grad.txt 2>&1
python3 - <<'PY'
import csv, json
O = "gpurun_out/r06_bwd/"
def stats(f):
    return {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(O + f))}
st, st1 = stats("r06_kernel_stats_backward.csv"), stats("r06_kernel_stats_backward_single.csv")
def find(d, pat):
    return next(v for k, v in d.items() if pat in k)
acc, dec = json.load(open(O + "accumulate.json")), json.load(open(O + "decoder.json"))
hbm = lambda d: 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
cycles = dec["GRBM_GUI_ACTIVE"] / 8                       # per XCD
rec = {"kernel": acc["kernel"], "views_per_launch": 4, "samples_per_launch": 4 * 128 * 128 * 96,
       "hbm_bytes_per_launch": hbm(acc), "avg_ns_profiled": find(st, "bwd_accumulate_reg")[0], "avg_ns_under_pmc": acc["avg_ns_profiled"],
       "decoder_kernel": {"kernel": dec["kernel"], "avg_ns_trace": find(st, "bwd_decoder_kernel")[0], "avg_ns_under_pmc": dec["avg_ns_profiled"],
                          "single_wave_kernel_avg_ns_trace": find(st1, "bwd_scatter_sorted_kernel<true, true>")[0],
                          "hbm_bytes_per_launch": hbm(dec), "cycles_under_pmc": cycles,
                          "valu_active": dec["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles, "ta_busy": dec["TA_TA_BUSY"] / 256 / cycles,
                          "mfma_busy": dec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cycles, "lds_issue": dec["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / cycles,
                          "valu_instructions_per_item": dec["SQ_INSTS_VALU"] / (4 * 128 * 128 * 96 / 64), "counters": dec},
       "other_kernels_us": {k.split("(")[0].replace("nfe::", ""): round(v[0] / 1e3, 1) for k, v in st.items()
                            if any(p in k for p in ("color_dot", "bin_fill", "bwd_ray", "bin_scan", "bwd_frag", "bwd_prep"))},
       "note": "tools/r06_backward_profile.sh: `BOTH_ONLY=1 tools/time_backward.py 4 128 48 48 256` on one box under rocprofv3 --kernel-trace "
               "--stats (avg_ns_profiled / avg_ns_trace) and under the --pmc passes of tools/pmc.sh (avg_ns_under_pmc, counters); FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md; fractions: unit-busy cycles / (units x GRBM_GUI_ACTIVE / 8 XCDs)"}
json.dump(rec, open(O + "r06_backward_counters.json", "w"), indent=1)
d = rec["decoder_kernel"]
print("decoder-backward", round(d["avg_ns_trace"] / 1e3, 1), "us (single-wave kernel", round(d["single_wave_kernel_avg_ns_trace"] / 1e3, 1), "us)  valu", round(d["valu_active"], 3),
      "ta", round(d["ta_busy"], 3), "mfma", round(d["mfma_busy"], 3), "lds", round(d["lds_issue"], 3), "VALU/item", round(d["valu_instructions_per_item"]))
print("accumulate", round(rec["avg_ns_profiled"] / 1e3, 1), "us", round(rec["hbm_bytes_per_launch"] / 1e9, 2), "GB", round(rec["hbm_bytes_per_launch"] / rec["avg_ns_profiled"] / 1e3, 2), "TB/s")
print(rec["other_kernels_us"])
PY
cat $OUT/time_backward.txt $OUT/time_sr_grad.txt
