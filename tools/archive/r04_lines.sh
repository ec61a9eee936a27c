#!/bin/bash
# Round 4: bench lines of every workload on one box (-> profiles/r04_bench_line*.json)
export TMPDIR=/tmp
OUT=gpurun_out/r04_lines
mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_line.json 2> $OUT/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-collective --no-cpu-baseline > $OUT/r04_bench_line_force_collective.json 2>> $OUT/bench.err
for w in full ffhq twopass editstep orbit; do
  python3 bench.py --workload $w --steps 10 --warmup 2 > $OUT/r04_bench_line_$w.json 2>> $OUT/bench.err
done
python3 bench.py --workload ffhq --conv-math fp16 --steps 10 --warmup 2 > $OUT/r04_bench_line_ffhq_fp16.json 2>> $OUT/bench.err
python3 bench.py --workload full --conv-math fp16 --steps 10 --warmup 2 > $OUT/r04_bench_line_full_fp16.json 2>> $OUT/bench.err
python3 bench.py --workload ffhq --conv-math bf16 --steps 10 --warmup 2 > $OUT/r04_bench_line_ffhq_bf16.json 2>> $OUT/bench.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_lines/r04_bench_line*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), d.get("strong_scaling", {}).get("views_per_s"), d["config"].get("stage_ms"), d["roofline"].get("bound"), d["roofline"].get("frac"), d.get("distributed"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/bench.err
