#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_fuzz
mkdir -p $OUT
python3 tools/fuzz_dense.py 11 400 2>&1 | grep -v "^/opt" | tail -8 > $OUT/fuzz_dense.txt
python3 tools/fuzz_dense.py 12 400 2>&1 | grep -v "^/opt" | tail -8 >> $OUT/fuzz_dense.txt
python3 tools/fuzz_sweep.py 500 400 150 2>&1 | grep -v "^/opt" | tail -6 > $OUT/fuzz_sweep.txt
python3 tools/soak_determinism.py 30 2>&1 | grep -v "^/opt" | tail -6 > $OUT/soak.txt
python3 tools/soak_synthesis.py 2>&1 | grep -v "^/opt" | tail -4 >> $OUT/soak.txt
cat $OUT/fuzz_dense.txt $OUT/fuzz_sweep.txt $OUT/soak.txt
