#!/usr/bin/env python3
"""Run-to-run reproducibility of nfe_render_backward on the editing-size case (the body of test_backward_is_repeatable), for
delta-debugging the lane-mask finding of profiles/experiments/r02_lane_mask.md on variant builds:
    bash tools/build_variant.sh tapsc -DNFE_TAPS_COMBINED=1
    NFE_RENDER_LIB=nerffaceediting_amd/csrc/build/variants/tapsc.so python tools/repro_lane_mask.py [repeats]
Prints, per repeat, the largest difference to the first run relative to the largest gradient entry and the number of differing
entries; exit code 1 if any run differs by more than 1e-6."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerffaceediting_amd import ops
from oracle import render_oracle as orc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
N, R, D, Di, H = 2, 128, 48, 48, 256
g = torch.Generator(device="cpu").manual_seed(3)
pn = torch.randn(N, 3, H, H, 32, generator=g).to(dev)
pd = (torch.randn(N, 3, H, H, 32, generator=g) * 1.3 + 0.2).to(dev)
shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
heads[3][0] += 2.0
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
c2w = np.concatenate([orc.lookat_pose(np.pi / 2 + y, np.pi / 2, [0, 0, 0.2], 2.7).reshape(1, 4, 4) for y in (-0.4, 0.4)])
K = np.stack([orc.fov_to_intrinsics(18.837)] * N)
kw = dict(cam2world=t(c2w), intrinsics=t(K), resolution=R)
opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)
cots = tuple(torch.randn(N, R * R, c, generator=g).to(dev) for c in (32, 15, 1, 1))
out = ops.render(pn, pd, ops.decoder_pack(*heads), opts, seed=1, taps=True, **kw)
first, worst = None, 0.0
for r in range(reps):
    gg, ga = ops.render_backward(pn, pd, heads, 1.0, opts, out[4]["depths_all"], cots, **kw)
    cur = (gg.clone(), ga.clone())
    if first is None:
        first = cur
        print(f"lib {os.environ.get('NFE_RENDER_LIB', 'shipped')}: largest entries {float(gg.abs().max()):.4g} / {float(ga.abs().max()):.4g}")
        continue
    for k, name in ((0, "geo"), (1, "app")):
        d = (cur[k] - first[k]).abs()
        rel = float(d.max()) / float(first[k].abs().max())
        worst = max(worst, rel)
        print(f"  run {r} {name}: max diff {rel:.3e} of the largest entry, {int((d > 1e-6 * first[k].abs().max()).sum())} entries differ")
print("REPRODUCED" if worst > 1e-6 else "repeatable")
sys.exit(1 if worst > 1e-6 else 0)
