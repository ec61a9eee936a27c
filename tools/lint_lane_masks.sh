#!/bin/bash
# Kept for muscle memory: the lint is tools/lint_lane_masks.py now (all four .hip files, e32 + e64 selects, any SGPR-pair
# destination, the uniform-branch shape reported beside it).  Writes the per-kernel table to profiles/ when asked:
#   tools/lint_lane_masks.sh [--report profiles/r03_lane_mask_lint.txt] [file.hip ...]
exec python3 "$(dirname "$0")/lint_lane_masks.py" "$@"
