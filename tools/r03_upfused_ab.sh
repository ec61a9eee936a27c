#!/bin/bash
# fused up-sampling epilogue: which layers (by Cin) and which tile height pay, per math mode
run() { echo "== $*"; env "$@" python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | cut -c1-120; env "$@" python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | cut -c1-120; }
run NFE_UP_FUSED=0
run NFE_UP_FUSED=1
run NFE_UP_FUSED_TALL=1
run NFE_UP_FUSED_TALL=1 NFE_UP_FUSED_CIN_X3=256
run NFE_UP_FUSED_CIN_X3=256
