#!/bin/bash
mkdir -p gpurun_out/r03_run3
python -m pytest tests -m gpu -q -x > gpurun_out/r03_run3/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_run3/rc.txt
tail -3 gpurun_out/r03_run3/pytest.log
python bench.py --workload full --steps 20 --warmup 3 > gpurun_out/r03_run3/bench_full.json 2>/dev/null
python bench.py --workload ffhq --steps 30 --warmup 5 > gpurun_out/r03_run3/bench_ffhq.json 2>/dev/null
python bench.py --workload editstep --steps 20 --warmup 3 > gpurun_out/r03_run3/bench_editstep.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-strong-scaling > gpurun_out/r03_run3/bench_default.json 2>/dev/null
for f in full ffhq editstep default; do python3 -c "
import json
d=json.loads(open('gpurun_out/r03_run3/bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), d['config'].get('stage_ms') or d['config'].get('forward_ms'), d['roofline'].get('frac'), d['roofline'].get('detail',{}).get('effective_clock_ghz'))"; done
