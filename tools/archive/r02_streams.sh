#!/bin/bash
# views/s of the full-synthesis workloads against the number of HIP streams the batches alternate on
for s in 1 2 3 4; do
  python3 bench.py --workload ffhq --steps 24 --warmup 4 --streams $s 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ffhq streams', $s, round(d['value'],1), round(d['ms_per_step'],3))"
done
for s in 2 3; do
  python3 bench.py --workload full --steps 12 --warmup 3 --streams $s 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('full streams', $s, round(d['value'],1), round(d['ms_per_step'],3))"
done
