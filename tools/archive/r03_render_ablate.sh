#!/bin/bash
V=nerffaceediting_amd/csrc/build/variants
mkdir -p gpurun_out/r03_render_ablate
for name in shipped noGather noMLP "$@" shipped; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  NFE_RENDER_LIB=$lib python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strong-scaling 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'fp32', round(d['roofline']['kernel_ms_fp32_exact'],3), 'Mrays/s', round(d['value']/1e6,1))"
done 2>&1 | tee gpurun_out/r03_render_ablate/ablate.log
