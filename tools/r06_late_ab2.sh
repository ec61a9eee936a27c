# Round 6: quarter-of-the-taps timing build against the shipped library (config 5 kernels, config 2 headline) and the PMC passes of the
# ideal-order launch.  build/variants/tapq.so is NOT in the tree: it was built with tools/build_variant.sh from a working copy whose
# NFE_WSI_ISSUE macro (nfe_render.hip) loads tap K = 0 only and fills the other taps' registers from the weight (wrong results, timing only);
# profiles/experiments/r06_render_floor.md section 2 has the numbers.
export TMPDIR=/tmp
OUT=gpurun_out/r06_late
mkdir -p $OUT
echo "== quarter-taps ablation (wrong results): config 5 kernels" | tee $OUT/tapq.txt
for lib in nerffaceediting_amd/libnfe_render.so nerffaceediting_amd/csrc/build/variants/tapq.so; do
  rm -rf $OUT/st
  NFE_RENDER_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 tools/cfg5_order.py merged 10 > $OUT/tapq_run.log 2>&1
  echo $lib | tee -a $OUT/tapq.txt
  python3 - $OUT/st >> $OUT/tapq.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_ws_kernel" in r["Name"] or "importance" in r["Name"]:
            print("   ", r["Name"][:80], "calls", r["Calls"], "avg_ns", r["AverageNs"])
PY
  tail -4 $OUT/tapq.txt
done
rm -rf $OUT/st
echo "== config 2 headline, same libs" | tee -a $OUT/tapq.txt
AB_STEPS=30 bash tools/ab.sh nerffaceediting_amd/libnfe_render.so nerffaceediting_amd/csrc/build/variants/tapq.so 2>&1 | tee -a $OUT/tapq.txt
PMC_TIMEOUT=200 PMC_PROG=tools/cfg5_order.py PMC_KERNEL="render_ws_kernel<4, 2, true, false, true, false>" bash tools/pmc.sh r06_late/pmc_ideal ideal 3 > $OUT/pmc_ideal.txt 2>&1
cp $OUT/pmc_ideal/issue_floor.json $OUT/r06_issue_floor_twopass_ideal_order.json
rm -rf $OUT/pmc_ideal/*/
tail -5 $OUT/pmc_ideal.txt | cut -c1-400
