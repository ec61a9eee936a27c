"""Debug: run the binned backward several times in one process and compare the intermediate buffers in the workspace
(feature-gradient rows, bin records, sorted list) between runs."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import ops  # noqa: E402

N, R, D, Di, H = 2, 128, 48, 48, 256
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
S = D + Di
ns = N * M * S
a256 = lambda x: (x + 255) & ~255
off = 32768 + 3 * a256(ns * 4) + a256((4 * 2048 + 64 + 64 + 32 + 32 + 8192) * 4) + a256(52 * 64 * 16)
slots = ns                      # one chunk (R*R is a multiple of 64)
sizes = [("df", slots * 256), ("rec_key", slots * 24), ("rec_w", slots * 48), ("binrank", slots * 24)]        # workspace order of nfe_render_backward
snaps = []
for rep in range(4):
    gg, ga = ops.render_backward(pn, pd, heads, 1.0, opts, out[4]["depths_all"], cots, cam2world=c2w, intrinsics=K, resolution=R)
    torch.cuda.synchronize()
    ws = list(ops._workspaces.values())[0]
    o = off
    snap = {}
    for name, nb in sizes:
        snap[name] = ws[o:o + nb].clone()
        o += a256(nb)
    snap["gg"] = gg.clone()
    snaps.append(snap)
for rep in range(1, 4):
    for name in ["df", "rec_key", "rec_w", "binrank", "gg"]:
        a, b = snaps[0][name], snaps[rep][name]
        if name == "df":
            x, y = a.view(torch.float32).view(-1, 64), b.view(torch.float32).view(-1, 64)
            bad = ((x - y).abs() > 1e-6 * x.abs().max()).any(dim=1).nonzero().flatten()
            print(f"run {rep} df rows differing: {len(bad)}", (bad[:10] // 64).tolist(), "lanes", (bad[:10] % 64).tolist())
            if len(bad):
                r = int(bad[0])
                print("   row", r, "run0", x[r, :4].tolist(), x[r, 32:36].tolist(), "runN", y[r, :4].tolist(), y[r, 32:36].tolist())
                waves = torch.unique(bad // 64)
                print("   waves", waves[:20].tolist(), "rows per wave", [(int((bad // 64 == w).sum())) for w in waves[:20]])
        elif name == "binrank":
            x, y = a.view(torch.int32).view(-1, 2), b.view(torch.int32).view(-1, 2)
            print(f"run {rep} bins differing: {int((x[:, 0] != y[:, 0]).sum())} (ranks may differ)")
        elif name == "gg":
            print(f"run {rep} grad max diff {(a - b).abs().max().item():.3e} of {a.abs().max().item():.3e}")
        else:
            print(f"run {rep} {name} bytes differing: {int((a != b).sum())}")
            if name == "rec_w":
                x, y = a.view(torch.float32).view(-1, 4), b.view(torch.float32).view(-1, 4)
                bad = (x != y).any(dim=1).nonzero().flatten()
                br = snaps[0]["binrank"].view(torch.int32).view(-1, 2)
                for r in bad[:24].tolist():
                    p_, idx = divmod(r, slots)
                    print(f"   slot {r}: plane {p_} wave {idx // 64} lane {idx % 64} bin {int(br[r, 0])}  run0 {x[r].tolist()}  runN {y[r].tolist()}")
                print("   waves:", torch.unique((bad % slots) // 64).tolist()[:40])
