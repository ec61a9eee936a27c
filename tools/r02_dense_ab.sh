#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_dense
mkdir -p $OUT
python3 -m pytest tests/test_e2e_gpu.py tests/test_dense_gpu.py -m gpu -x -q 2>&1 | tail -3 > $OUT/tests_streams.txt
cat $OUT/tests_streams.txt
for st in 1 2 3; do
python3 bench.py --workload ffhq --steps 20 --warmup 4 --streams $st 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ffhq streams', d['config']['streams'], d['value'], d['ms_per_step'], d['config']['stage_ms'])"
done
python3 bench.py --workload full --steps 10 --warmup 2 --streams 2 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('full', d['value'], d['ms_per_step'])"
python3 bench.py --workload orbit --steps 2 --warmup 1 --streams 2 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('orbit', d['value'], d['ms_per_step'])"
