"""Hash of the render outputs over the kernel variants, for one value of NFE_RENDER_BLOCKS_PER_CU / one library
(NFE_RENDER_LIB).  Equal hashes across occupancies = no dependence on what shares the CU (tests/test_render_gpu.py runs the
same cases inside the suite; this stand-alone form is for experimental builds).
    NFE_RENDER_BLOCKS_PER_CU=4 NFE_RENDER_LIB=.../variant.so python tools/hash_occupancy.py [repeats]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerffaceediting_amd import ops  # noqa: E402
from oracle import render_oracle as orc  # noqa: E402  (camera construction only)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
N, H = 2, 256
raw = torch.randn(N, 96, H, H, generator=g).to(dev)
mean, std = ops.plane_stats(raw)
packed = ops.plane_pack(raw)
aff = ops.make_affine(mean, std)
packed2 = ops.plane_pack((raw * 0.8 + 0.1).contiguous())
rawr = torch.randn(N, 96, 192, 320, generator=g).to(dev)
packedr = ops.plane_pack(rawr)
affr = ops.make_affine(*ops.plane_stats(rawr))
shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
dec = ops.decoder_pack(*[(torch.randn(*s, generator=g) * (1.0 if len(s) == 2 else 0.2)).to(dev) for s in shapes])
c2w = torch.from_numpy(np.concatenate([orc.lookat_pose(np.pi / 2 + y, np.pi / 2 + p, [0, 0, 0.2], 2.7).reshape(1, 4, 4) for y, p in ((0.3, -0.2), (-0.9, 0.4))])).to(dev)
K = torch.from_numpy(np.repeat(orc.fov_to_intrinsics(18.837)[None], N, 0)).to(dev)
cases = [(packed, packed, aff, 512, 64, 0, None), (packed, packed, aff, 512, 24, 24, None),
         (packedr, packedr, affr, 512, 64, 0, None), (packedr, packedr, affr, 256, 24, 24, None),
         (packed, packed2, None, 512, 64, 0, None), (packed, packed2, None, 256, 96, 96, None),
         (packed, packed, aff, 256, 96, 96, None), (packed, packed, aff, 256, 48, 48, "fp32"), (packed, packed2, None, 256, 32, 0, "fp32")]
if os.environ.get("HASH_SQUARE_ONLY"):          # experimental builds that hard-wire the square path must not see other planes
    cases = [c for c in cases if c[0] is packed and c[1] is packed and c[6] is None]
for pg, pa, af, R, D, Di, math in cases:
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)
    hs = set()
    for _ in range(reps):
        out = ops.render(pg, pa, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, affines=af, seed=5, decoder_math=math)
        hh = hashlib.sha256()
        for t in out:
            hh.update(t.cpu().numpy().tobytes())
        hs.add(hh.hexdigest()[:16])
    print("CASE", R, D, Di, math, tuple(pg.shape[2:4]), "same" if pg is pa else "dual", " ".join(sorted(hs)), "UNSTABLE" if len(hs) > 1 else "")
