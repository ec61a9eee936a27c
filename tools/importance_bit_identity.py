import hashlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from nerffaceediting_amd import ops
dev = torch.device("cuda:0")
planes, dec_t, _, c2w_t, K_t, _, _, _ = bench.synth_inputs(torch, dev, 1000)
mean, std = ops.plane_stats(planes)
aff = ops.make_affine(mean, std)
packed = ops.plane_pack(planes)
names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias", "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
dec = ops.decoder_pack(*[dec_t[k] for k in names])
h = hashlib.sha256()
for R, D, Di in ((128, 48, 48), (256, 96, 96), (64, 200, 256), (64, 64, 64), (64, 128, 128), (64, 17, 5)):
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False, clamp_mode="softplus")
    out = ops.render(packed, packed, dec, opts, cam2world=c2w_t, intrinsics=K_t, resolution=R, seed=77, affines=aff, taps=True)
    hh = hashlib.sha256()
    for t in list(out[:4]) + [out[4]["depths_all"], out[4]["depths_fine"]]:
        hh.update(t.cpu().numpy().tobytes())
    print(R, D, Di, hh.hexdigest()[:16])
    h.update(hh.digest())
print("HASH", h.hexdigest())
