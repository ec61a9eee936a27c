#!/bin/bash
# round 3, GPU call 1: whole -m gpu suite (prints kept), default bench line, dense workloads as the round's baseline
mkdir -p gpurun_out/r03_run1
python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_run1/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_run1/rc.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_run1/bench_default.json 2> gpurun_out/r03_run1/bench_default.err; echo "bench rc=$?" >> gpurun_out/r03_run1/rc.txt
python bench.py --workload ffhq --steps 30 --warmup 5 > gpurun_out/r03_run1/bench_ffhq.json 2>> gpurun_out/r03_run1/bench_default.err
python bench.py --workload full --steps 20 --warmup 3 > gpurun_out/r03_run1/bench_full.json 2>> gpurun_out/r03_run1/bench_default.err
tail -5 gpurun_out/r03_run1/pytest.log; cat gpurun_out/r03_run1/rc.txt
