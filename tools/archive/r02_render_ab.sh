#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_render
mkdir -p $OUT
python3 -m pytest tests/test_render_gpu.py tests/test_render_edge_gpu.py tests/test_renderer_interface_gpu.py tests/test_random_sweep_gpu.py tests/test_render_backward_gpu.py -m gpu -x -q 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-scaling 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('render', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_fp32_exact'])"; done
python3 bench.py --workload twopass --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('twopass', d['value'], d['ms_per_step'])"
python3 bench.py --workload ffhq --steps 20 --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('ffhq', d['value'], d['ms_per_step'])"
