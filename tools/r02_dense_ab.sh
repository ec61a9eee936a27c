#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_dense
mkdir -p $OUT
python3 -m pytest tests/test_dense_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tail -3
python3 tools/fuzz_dense.py 21 300 2>&1 | grep -v "^/opt" | tail -3
for k in 0 1 0 1; do
NFE_C3_KSPLIT=$k python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s/^/ksplit=$k /"
NFE_C3_KSPLIT=$k python3 tools/time_full.py 1 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s/^/ksplit=$k /"
done
python3 bench.py --workload ffhq --steps 20 --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ffhq', d['value'], d['ms_per_step'])"
