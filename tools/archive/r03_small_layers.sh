#!/bin/bash
# 4^2 - 16^2 layers on the conv3 fast path (32-wide tiles, mostly padding) instead of the generic kernel?
run() { echo "== $*"; env "$@" python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | cut -c1-120; env "$@" python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | cut -c1-120; env "$@" python3 tools/time_full.py 1 128 48 48 bf16x3 2>&1 | grep -E "^N=" | cut -c1-120; }
run NFE_C3_MIN_W=32
run NFE_C3_MIN_W=16
run NFE_C3_MIN_W=8 NFE_C3_MIN_H=8
run NFE_C3_MIN_W=4 NFE_C3_MIN_H=4
