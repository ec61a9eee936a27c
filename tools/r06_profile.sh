#!/bin/bash
# Round 6: every number the bench lines and DESIGN.md quote, from ONE box and ONE build.
#   1. PMC passes (tools/pmc.sh: five rocprofv3 --pmc passes of the command itself) of the headline command (render_ws_kernel and the
#      exact-fp32 render_kernel both run in it), of --workload twopass (three kernels) and of the dense path (tools/time_full.py at the
#      config-3 shape in bf16, the FFHQ shape in fp16 and in split-bf16)
#   2. rocprofv3 --kernel-trace --stats of the headline, twopass, full and ffhq commands
#   3. the bench line of every workload
#   4. tools/r06_backward_profile.sh (gpurun_out/r06_bwd/)
# Output: gpurun_out/r06_profile/ (copy what is quoted into profiles/).
export TMPDIR=/tmp
export PMC_TIMEOUT=${PMC_TIMEOUT:-150}
OUT=gpurun_out/r06_profile
mkdir -p $OUT
pick() { PMC_KERNEL="$1" python3 tools/pmc_summary.py $OUT/$2 > $OUT/$3.txt 2>&1; cp $OUT/$2/issue_floor.json $OUT/$4; }
bash tools/pmc.sh r06_profile/pmc > $OUT/pmc_default.log 2>&1
pick "render_ws_kernel<4, 2, true, false, false, false>" pmc r06_pmc_render_ws r06_issue_floor.json
pick "render_kernel<false, false, 1," pmc r06_pmc_render_fp32 r06_issue_floor_fp32.json
bash tools/pmc.sh r06_profile/pmc2 --workload twopass --steps 3 --warmup 1 > $OUT/pmc_twopass.log 2>&1
pick "render_ws_kernel<4, 2, true, true, true, false>" pmc2 r06_pmc_twopass r06_issue_floor_twopass_final.json
pick "render_ws_kernel<4, 2, true, false, false, true>" pmc2 r06_pmc_twopass_sigma r06_issue_floor_twopass_sigma.json
pick "importance_kernel" pmc2 r06_pmc_twopass_importance r06_issue_floor_twopass_importance.json
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<1, 2, false, 2, 8, 2" bash tools/pmc.sh r06_profile/pmc_bf16 8 512 64 0 bf16 > $OUT/r06_pmc_dense_bf16.txt 2>&1
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<2, 2, false, 2, 8, 2" bash tools/pmc.sh r06_profile/pmc_fp16 4 128 48 48 fp16 > $OUT/r06_pmc_dense_fp16.txt 2>&1
PMC_PROG=tools/time_full.py PMC_KERNEL="conv3_kernel<3, 2, false, 1, 4, 4" bash tools/pmc.sh r06_profile/pmc_x3 4 128 48 48 bf16x3 > $OUT/r06_pmc_dense_x3.txt 2>&1
rm -rf $OUT/pmc/*/ $OUT/pmc2/*/ $OUT/pmc_bf16/*/ $OUT/pmc_fp16/*/ $OUT/pmc_x3/*/
python3 tools/dense_pmc_table.py $OUT/r06_dense_kernels.json bf16=$OUT/r06_pmc_dense_bf16.txt fp16=$OUT/r06_pmc_dense_fp16.txt bf16x3=$OUT/r06_pmc_dense_x3.txt > /dev/null 2>&1 || true
stats() { timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" --no-cpu-baseline --no-strong-scaling > $OUT/stats.log 2>&1; find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/$NAME \; ; rm -rf $OUT/stats; }
NAME=r06_kernel_stats.csv stats --steps 20 --warmup 5
NAME=r06_kernel_stats_twopass.csv stats --workload twopass --steps 10 --warmup 2
NAME=r06_kernel_stats_dense_full.csv stats --workload full --steps 10 --warmup 2
NAME=r06_kernel_stats_dense_ffhq.csv stats --workload ffhq --steps 10 --warmup 2
NAME=r06_kernel_stats_dense_ffhq_fp16.csv stats --workload ffhq --conv-math fp16 --steps 10 --warmup 2
python3 bench.py --gpus 1 > $OUT/r06_bench_line.json 2> $OUT/bench.err
python3 bench.py --gpus 1 --force-collective --no-cpu-baseline > $OUT/r06_bench_line_force_collective.json 2>> $OUT/bench.err
for w in full ffhq twopass editstep orbit; do
  python3 bench.py --workload $w --steps 10 --warmup 2 > $OUT/r06_bench_line_$w.json 2>> $OUT/bench.err
done
python3 bench.py --workload ffhq --conv-math fp16 --steps 10 --warmup 2 > $OUT/r06_bench_line_ffhq_fp16.json 2>> $OUT/bench.err
python3 bench.py --workload full --conv-math fp16 --steps 10 --warmup 2 > $OUT/r06_bench_line_full_fp16.json 2>> $OUT/bench.err
python3 bench.py --workload ffhq --conv-math bf16 --steps 10 --warmup 2 > $OUT/r06_bench_line_ffhq_bf16.json 2>> $OUT/bench.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_profile/r06_bench_line*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), d.get("strong_scaling", {}).get("views_per_s"), d["config"].get("stage_ms"), d["roofline"].get("bound"), d["roofline"].get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h '"kernel"\|avg_ns_profiled' $OUT/r06_issue_floor*.json
head -6 $OUT/r06_kernel_stats.csv | cut -c1-180
tail -3 $OUT/bench.err
# 4. the edit step's backward (kernel trace + PMC passes of tools/time_backward.py, both decoder-backward kernels) and the SR-head gradient
timeout 900 bash tools/r06_backward_profile.sh
