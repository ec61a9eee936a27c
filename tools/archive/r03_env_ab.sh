#!/bin/bash
# round 3: A/B of run-time switches (environment variables) of the shipped library.  tools/r03_env_ab.sh "VAR=1" "VAR2=1 VAR3=0" ...
export TMPDIR=/tmp
OUT=gpurun_out/r03_env_ab
mkdir -p $OUT
for setting in "" "$@" ""; do
  tag=${setting:-shipped}
  if [ -n "$setting" ]; then
    env $setting python3 -m pytest tests/test_dense_gpu.py tests/test_batched_dense_gpu.py -m gpu -x -q -k "not repeated and not orbit" 2>&1 | tail -1 | sed "s|^|[$tag] parity: |"
  fi
  for rep in 1 2; do
    env $setting python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | sed "s|^|[$tag] |"
  done
  env $setting python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | sed "s|^|[$tag] |"
done 2>&1 | tee $OUT/ab_$(date +%H%M%S).log
