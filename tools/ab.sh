#!/bin/bash
# A/B timing of render-kernel builds on the GPU box: tools/ab.sh lib1.so lib2.so ...  (3 interleaved repetitions)
for rep in 1 2 3; do
  for lib in "$@"; do
    ms=$(NFE_RENDER_LIB=$PWD/$lib python3 bench.py --steps ${AB_STEPS:-20} --warmup 3 --no-cpu-baseline --no-strong-scaling 2>/dev/null | tail -1 | grep -oE '"kernel_(ms|mcycles)": [0-9.]+' | cut -d' ' -f2 | tr '\n' ' ')
    echo "$lib $ms"
  done
done
