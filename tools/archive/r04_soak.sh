#!/bin/bash
# Round 4, after the packed-fp32 operand fix: every kernel variant hashed over repeated launches at 1 / 2 / 4 workgroups per CU (fused and
# wave-specialised launch), the backward repeated 30 times, the 500-launch three-stream soak twice.  Any UNSTABLE / differing line = a failure.
export TMPDIR=/tmp
for ws in 42 0; do for b in 1 2 4; do
  echo "== render hashes: NFE_RENDER_WS=$ws, $b workgroup(s) per CU, 12 launches per case"
  NFE_RENDER_WS=$ws NFE_RENDER_BLOCKS_PER_CU=$b timeout 600 python3 tools/hash_occupancy.py 12 2>&1 | grep CASE
done; done
echo "== backward, 30 launches"; timeout 300 python3 tools/repro_lane_mask.py 30 2>&1 | tail -3
echo "== soak tests"; for i in 1 2; do timeout 900 python3 -m pytest tests/test_batched_dense_gpu.py -q -k soak 2>&1 | tail -1; done
timeout 600 python3 -m pytest tests/test_render_backward_gpu.py -q -k "repeatable or survives" 2>&1 | tail -1
