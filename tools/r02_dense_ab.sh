#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_dense
mkdir -p $OUT
python3 -m pytest tests/test_dense_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tail -3 > $OUT/tests_ep.txt
cat $OUT/tests_ep.txt
NFE_RENDER_LIB=nerffaceediting_amd/csrc/build/variants/c3prof.so python3 tools/c3_profile.py 2>&1 | grep -v "^/opt" | tee $OUT/c3_profile2.txt
python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N="
python3 tools/time_full.py 8 512 64 0 bf16 2>&1 | grep -E "^N="
python3 bench.py --workload ffhq --steps 20 --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ffhq', d['value'], d['ms_per_step'])"
