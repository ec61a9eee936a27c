#!/bin/bash
# A/B of the two-pass dual-plane workload over library builds: tools/ab_twopass.sh lib1.so lib2.so ...
for rep in 1 2; do
  for lib in "$@"; do
    NFE_RENDER_LIB=$PWD/$lib python3 bench.py --workload twopass --steps 4 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],2))"
  done
done
