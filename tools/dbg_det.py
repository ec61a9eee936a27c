import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerffaceediting_amd import ops
from oracle import render_oracle as orc
dev=torch.device("cuda:0")
rng=np.random.RandomState(7)
R=int(sys.argv[1]) if len(sys.argv)>1 else 512
planes=torch.from_numpy((rng.randn(1,96,256,256)*1.0).astype(np.float32)).to(dev)
dec=orc.random_decoder(9, bias_scale=0.1)
names=["geo_net.0.weight","geo_net.0.bias","geo_net.2.weight","geo_net.2.bias","app_net.0.weight","app_net.0.bias","app_net.2.weight","app_net.2.bias"]
decp=ops.decoder_pack(*[torch.from_numpy(dec[k]).to(dev) for k in names])
c2w=orc.lookat_pose(math.pi/2+0.4, math.pi/2-0.2,[0,0,0.2],2.7); K=orc.fov_to_intrinsics(18.837)[None]
mean,std=ops.plane_stats(planes); aff=ops.make_affine(mean,std); packed=ops.plane_pack(planes)
opts=dict(depth_resolution=64, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1)
kw=dict(cam2world=torch.from_numpy(c2w).to(dev), intrinsics=torch.from_numpy(K).to(dev), resolution=R, affines=aff)
outs=[ops.render(packed,packed,decp,opts,seed=5,**kw) for _ in range(3)]
torch.cuda.synchronize()
for i in (1,2):
    d=(outs[0][0]-outs[i][0]).abs().amax(-1)[0]
    bad=(d>0).nonzero().flatten()
    print("run",i,"mismatching rays:",bad.numel(),"max diff",float(d.max()))
    if bad.numel():
        ys=(bad//R).cpu().numpy(); xs=(bad%R).cpu().numpy()
        print(" first rays (y,x):",list(zip(ys[:12],xs[:12])))
        print(" tiles (y//4,x//8) distinct:",len(set(zip(ys//4,xs//8))))
import collections
d=(outs[0][0]-outs[1][0]).abs().amax(-1)[0]
bad=(d>0).nonzero().flatten().cpu().numpy()
ys, xs = bad//R, bad%R
rbs = sorted(set(((ys//4)*(R//8) + xs//8).tolist()))
print("bad ray blocks:", len(rbs), "wave slots:", collections.Counter(rb % 8 for rb in rbs), "first:", rbs[:16])
print("block ids:", sorted(set((rb//8) % 256 for rb in rbs))[:24])
lanes = collections.Counter(((y%4)*8 + x%8) for y,x in zip(ys.tolist(), xs.tolist()))
print("lanes j:", sorted(lanes.items())[:40])
