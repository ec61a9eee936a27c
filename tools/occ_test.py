import sys, os, time, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from nerffaceediting_amd import ops
from oracle import render_oracle as orc
dev=torch.device("cuda:0")
g=torch.Generator().manual_seed(0)
planes=torch.randn(4,96,256,256,generator=g).to(dev)
dec=orc.random_decoder(1)
names=["geo_net.0.weight","geo_net.0.bias","geo_net.2.weight","geo_net.2.bias","app_net.0.weight","app_net.0.bias","app_net.2.weight","app_net.2.bias"]
decp=ops.decoder_pack(*[torch.from_numpy(dec[k]).to(dev) for k in names])
c2w=np.concatenate([orc.lookat_pose(math.pi/2+y, math.pi/2-0.2,[0,0,0.2],2.7) for y in (0.4,0,-0.4,0.2)],0)
K=np.tile(orc.fov_to_intrinsics(18.837)[None],(4,1,1))
mean,std=ops.plane_stats(planes); aff=ops.make_affine(mean,std); packed=ops.plane_pack(planes)
opts=dict(depth_resolution=64, depth_resolution_importance=8, ray_start=2.25, ray_end=3.3, box_warp=1)
kw=dict(cam2world=torch.from_numpy(c2w).to(dev), intrinsics=torch.from_numpy(K.astype(np.float32)).to(dev), resolution=512, affines=aff)
for i in range(3): ops.render(packed,packed,decp,opts,seed=i,**kw)
torch.cuda.synchronize()
