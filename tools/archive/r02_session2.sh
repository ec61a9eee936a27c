#!/bin/bash
# GPU session 2: bisect of the run-time SQUARE-branch nondeterminism (profiles/experiments/r02_square_branch.md)
export TMPDIR=/tmp
OUT=gpurun_out/r02_s2
mkdir -p $OUT
V=nerffaceediting_amd/csrc/build/variants
for n in 2 3 4 5; do
  echo "== variant square_rt$n blocks_per_cu=2" >> $OUT/hash.txt
  NFE_RENDER_LIB=$V/square_rt$n.so NFE_RENDER_BLOCKS_PER_CU=2 python3 tools/hash_occupancy.py 4 2>&1 | grep -E "CASE (512 64 0|512 24 24|256 96 96) None \(256, 256\) same" >> $OUT/hash.txt
done
echo "== parity of square_rt at 1 block per CU" >> $OUT/hash.txt
NFE_RENDER_LIB=$V/square_rt.so NFE_RENDER_BLOCKS_PER_CU=1 python3 -m pytest tests/test_render_gpu.py -q -x -k "reference_golden or full_size_vs_reference or two_pass_dual" 2>&1 | tail -3 >> $OUT/hash.txt
cat $OUT/hash.txt
