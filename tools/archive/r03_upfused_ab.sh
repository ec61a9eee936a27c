#!/bin/bash
# fused up-sampling epilogue: which layers (by Cin) pay, per math mode (re-run after upfir_kernel got faster)
run() { echo "== $*"; env "$@" python3 tools/time_full.py 8 128 64 0 bf16 2>&1 | grep -E "^N=" | cut -c1-120; env "$@" python3 tools/time_full.py 4 128 48 48 bf16x3 2>&1 | grep -E "^N=" | cut -c1-120; }
run NFE_UP_FUSED=0
run NFE_UP_FUSED_CIN_BF16=32 NFE_UP_FUSED_CIN_X3=16
run NFE_UP_FUSED_CIN_BF16=128 NFE_UP_FUSED_CIN_X3=32
run NFE_UP_FUSED_CIN_BF16=256 NFE_UP_FUSED_CIN_X3=256
run NFE_UP_FUSED_CIN_BF16=512 NFE_UP_FUSED_CIN_X3=512
