# Round 6: the FFHQ line with the shipped launch rule, with render_ws_kernel<4,2> forced at 2 048 ray blocks, and with a variant build
# (build/variants/np2.so, not in the tree: launch_render_ws<2, 2> behind NFE_RENDER_WS_NP2=1 in launch_render, built with tools/build_variant.sh)
# - profiles/experiments/r06_ab_ffhq_ws_geometry.txt.
V=$PWD/nerffaceediting_amd/csrc/build/variants/np2.so
run() { python3 bench.py --workload ffhq --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline'].get('stage_ms') or d['config'].get('stage_ms'))"; }
for rep in 1 2 3; do
  echo "shipped            $(run)"
  echo "ws4 at ffhq        $(NFE_RENDER_LIB=$V NFE_RENDER_WS_MIN_RB=1024 run)"
  echo "ws np2 at ffhq     $(NFE_RENDER_LIB=$V NFE_RENDER_WS_MIN_RB=1024 NFE_RENDER_WS_NP2=1 run)"
done
