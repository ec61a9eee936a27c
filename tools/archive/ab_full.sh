#!/bin/bash
# A/B of the full-synthesis workload over library builds: tools/ab_full.sh lib1.so lib2.so ...
for rep in 1 2; do
  for lib in "$@"; do
    NFE_RENDER_LIB=$PWD/$lib python3 bench.py --workload full --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), {k: round(v,2) for k,v in d['config']['stage_ms'].items()})"
  done
done
