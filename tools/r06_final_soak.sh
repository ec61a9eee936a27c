#!/bin/bash
# Round 6, final build: determinism soaks and wide fuzz runs in one log (gpurun_out/r06_final_soak.txt)
OUT=gpurun_out/r06_final_soak.txt
: > $OUT
run() { echo "== $*" | tee -a $OUT; timeout 900 "$@" 2>&1 | grep -v "amdgpu.ids" | tail -12 | tee -a $OUT; echo "rc=$?" >> $OUT; }
run python3 tools/soak_determinism.py 40
run python3 tools/soak_lds_garbage.py 20
run python3 tools/soak_synthesis.py 10
run python3 tools/fuzz_dense.py 606 250
run python3 tools/fuzz_sweep.py 600 300 60
