#!/bin/bash
# conv3 fragment-read pipeline A/B (variant c3nopipe = -DC3_FRAG_PIPE=0): parity, fuzz, layer timings, bench lines
export TMPDIR=/tmp
OUT=gpurun_out/r02_dense
mkdir -p $OUT
python3 -m pytest tests/test_dense_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tail -3
python3 tools/fuzz_dense.py 21 300 2>&1 | grep -v "^/opt" | tail -3
V=nerffaceediting_amd/csrc/build/variants
for lib in "" $V/c3nopipe.so "" $V/c3nopipe.so; do
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|lib=${lib:-shipped} |"
  NFE_RENDER_LIB=$lib python3 tools/time_full.py 1 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|lib=${lib:-shipped} |"
done
for lib in "" $V/c3nopipe.so; do
  NFE_RENDER_LIB=$lib python3 bench.py --workload ffhq --steps 20 --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ffhq', d['value'], d['ms_per_step'])"
  NFE_RENDER_LIB=$lib python3 bench.py --workload full --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('full', d['value'], d['ms_per_step'])"
done
