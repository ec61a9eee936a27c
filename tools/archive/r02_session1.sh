#!/bin/bash
# GPU session 1 of round 2: VALU issue-rate microbenchmark (+ PMC calibration of the SQ counters on it), the run-time
# SQUARE-branch repro, and the new tests.   gpurun -- 'bash tools/r02_session1.sh'
export TMPDIR=/tmp
OUT=gpurun_out/r02_s1
mkdir -p $OUT
./tools/microbench/valu_rate > $OUT/valu_rate.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_valu -- ./tools/microbench/valu_rate > $OUT/pmc_valu.log 2>&1
python3 - <<'PY' > gpurun_out/r02_s1/pmc_valu_summary.txt 2>&1
import csv, glob, collections
f = glob.glob("gpurun_out/r02_s1/pmc_valu/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r["Workgroup_Size"] if "Workgroup_Size" in r else "")
    rows.setdefault(k, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in rows.items():
    print(k, {c: f"{x:.4g}" for c, x in sorted(v.items())})
PY
for lib in "" nerffaceediting_amd/csrc/build/variants/square_rt.so; do
  for b in 1 2 4; do
    echo "== lib=${lib:-shipped} blocks_per_cu=$b" >> $OUT/hash.txt
    NFE_RENDER_LIB=$lib NFE_RENDER_BLOCKS_PER_CU=$b python3 tools/hash_occupancy.py 4 >> $OUT/hash.txt 2>&1
  done
done
python3 -m pytest tests/test_renderer_interface_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/tests.txt
tail -3 $OUT/tests.txt
