#!/bin/bash
# Full GPU suite + the bench lines of every workload on one box (copied into profiles/ afterwards).
export TMPDIR=/tmp
OUT=gpurun_out/r02_lines
mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r02_bench_line.json 2> $OUT/bench.err
for w in full ffhq twopass editstep orbit; do
  python3 bench.py --workload $w --steps 10 --warmup 2 > $OUT/r02_bench_line_$w.json 2>> $OUT/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_lines/r02_bench_line*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["unit"], "ms/step", round(d["ms_per_step"], 3), d.get("strong_scaling", {}).get("views_per_s"), d["config"].get("stage_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $OUT/bench.err
