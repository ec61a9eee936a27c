#!/bin/bash
# Round 6: the split-bf16 convolutions' LDS-DMA path on layers narrower than 32 pixels (NFE_C3_MIN_W / NFE_C3_MIN_H are A/B knobs of
# conv3_eligible): backbone time of the FFHQ configuration by tools/time_full.py, two interleaved repetitions
for rep in 1 2; do
  for cfg in "0 0" "16 8" "8 8" "8 4" "4 4"; do
    set -- $cfg
    echo "min_w=$1 min_h=$2: $(NFE_C3_MIN_W=$1 NFE_C3_MIN_H=$2 python3 tools/time_full.py 4 128 48 48 bf16x3 2>/dev/null | head -1 | cut -c1-150)"
  done
done
