#!/bin/bash
# Extended fuzz / soak pass (other seeds than tools/r02_fuzz.sh)
export TMPDIR=/tmp
OUT=gpurun_out/r02_fuzz_long
mkdir -p $OUT
for s in 101 102 103; do python3 tools/fuzz_dense.py $s 400 2>&1 | grep -v "^/opt" | tail -1; done | tee $OUT/fuzz_dense.txt
python3 tools/fuzz_sweep.py 7000 1000 400 2>&1 | grep -v "^/opt" | tail -2 | tee $OUT/fuzz_sweep.txt
python3 tools/soak_determinism.py 50 2>&1 | grep -v "^/opt" | tail -6 | tee $OUT/soak.txt
python3 tools/cmp_bwd_forms.py 6 2>&1 | grep "binned" | tee $OUT/bwd_repeat.txt
