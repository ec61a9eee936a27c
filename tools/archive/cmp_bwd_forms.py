"""Compare the scatter forms of nfe_render_backward on one editing-size case: each form runs in a child interpreter (the switch is
read once per process), gradients are compared texel by texel.    python tools/cmp_bwd_forms.py [repeats [N R D Di H]]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nerffaceediting_amd import ops
N, R, D, Di, H = [int(v) for v in sys.argv[3:8]]
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
pn = torch.randn(N, 3, H, H, 32, generator=g).to(dev)
pd = (torch.randn(N, 3, H, H, 32, generator=g) * 1.3 + 0.2).to(dev)
shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
heads[3][0] += 2.0
dec = ops.decoder_pack(*heads)
th = torch.linspace(-0.4, 0.4, N)
c2w = torch.eye(4).repeat(N, 1, 1)
c2w[:, 0, 0], c2w[:, 0, 2], c2w[:, 2, 0], c2w[:, 2, 2] = torch.cos(th), torch.sin(th), -torch.sin(th), torch.cos(th)
c2w[:, :3, 2] *= -1
c2w[:, :3, 3] = -2.7 * c2w[:, :3, 2]
K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1)
c2w, K = c2w.to(dev), K.to(dev)
opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)
M = R * R
cots = tuple(torch.randn(N, M, c, generator=g).to(dev) for c in (32, 15, 1, 1))
out = ops.render(pn, pd, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, seed=1, taps=True)
for rep in range(int(sys.argv[2])):
    gg, ga = ops.render_backward(pn, pd, heads, 1.0, opts, out[4]["depths_all"], cots, cam2world=c2w, intrinsics=K, resolution=R)
    np.save(sys.argv[1] + "_g%%d.npy" %% rep, gg.cpu().numpy()); np.save(sys.argv[1] + "_a%%d.npy" %% rep, ga.cpu().numpy())
''' % ROOT


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    shape = sys.argv[2:7] if len(sys.argv) >= 7 else ["2", "128", "48", "48", "256"]        # N R D Di H
    for form, env in (("sorted", {"NFE_BWD_SCATTER": "sorted"}), ("binned", {})):
        subprocess.run([sys.executable, "-c", CHILD, "/tmp/cmp_" + form, str(reps)] + shape, env=dict(os.environ, **env), check=True)
    for which in "ga":
        ref = np.load(f"/tmp/cmp_sorted_{which}0.npy").astype(np.float64)
        scale = np.abs(ref).max()
        for form in ("sorted", "binned"):
            for rep in range(reps):
                x = np.load(f"/tmp/cmp_{form}_{which}{rep}.npy").astype(np.float64)
                d = np.abs(x - ref)
                bad = np.argwhere(d > 2e-4 * scale)
                print(f"{which} {form}[{rep}] vs sorted[0]: max {d.max() / scale:.2e} of scale, {len(bad)} entries over 2e-4")
                for b in bad[:12]:
                    n, p, y, xx, c = b
                    print("     view", n, "plane", p, "y", y, "x", xx, "ch", c, "got", x[tuple(b)], "want", ref[tuple(b)])


if __name__ == "__main__":
    main()
