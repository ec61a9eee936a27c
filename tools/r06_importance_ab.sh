# Round 6: parity tests + same-box A/B of importance_kernel: build/variants/base.so = the library before commit c11be5a (build it with
# tools/build_variant.sh from `git show c11be5a~1:nerffaceediting_amd/csrc/nfe_render.hip`), against the shipped one; tools/cfg5_order.py merged.
export TMPDIR=/tmp
OUT=gpurun_out/r06_late
mkdir -p $OUT
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_renderer_interface_gpu.py -x -q 2>&1 | tail -3
for lib in nerffaceediting_amd/csrc/build/variants/base.so nerffaceediting_amd/libnfe_render.so nerffaceediting_amd/csrc/build/variants/base.so nerffaceediting_amd/libnfe_render.so; do
  rm -rf $OUT/st
  NFE_RENDER_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 tools/cfg5_order.py merged 10 > $OUT/imp_run.log 2>&1
  echo "$lib $(tail -1 $OUT/imp_run.log | cut -c1-120)"
  python3 - $OUT/st <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "importance" in r["Name"]:
            print("   ", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"])
PY
done
rm -rf $OUT/st
