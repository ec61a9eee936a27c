"""How many distinct texels does an 8 x 4 ray tile touch per plane and depth step?  (test infrastructure: CPU only, never imported by the product)

north_star's "LDS staging of per-tile plane patches": a wave step of the render kernels evaluates ONE sample index of 32 neighbouring
rays; if their 4 x 32 taps on a plane fall on few distinct texels, the tile's texels could be staged in LDS once per step and the taps
served from there (4-5 x fewer texture requests).  This script measures the footprint on the scene of the config-5 fixture
(tests/golden/cfg5_render_ws.npz: 512^2 rays, 96 + 96 samples, 256^2 planes) with the oracle's own depths:

    python oracle/texel_window_census.py [n_tiles]

for the COARSE pass (stratified depths: all rays of a tile are within one stratum of each other) and for the FINAL pass (merged depth
lists: at one list index the rays of a tile sit at unrelated depths, because every ray's importance samples cluster at its own surface).
Planes: p0 = (x, y), p1 = (x, z), p2 = (z, x) (renderer.py:39-53).  Result, round 5 (256 tiles): see profiles/experiments/r05_render_negative.md.
"""
import ast
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import render_oracle as orc  # noqa: E402


def footprint(coords, H, W):
    """coords [T, 32, 3] in [-1, 1] (already / box_warp * 2): per plane the number of distinct texels under the 4 bilinear taps of the
    tile's 32 rays, and the area of their bounding box -> ([T, 3], [T, 3])."""
    proj = [(0, 1), (0, 2), (2, 0)]
    T = coords.shape[0]
    distinct = np.zeros((T, 3), np.int64)
    bbox = np.zeros((T, 3), np.int64)
    for p, (a, b) in enumerate(proj):
        ix = np.floor((coords[..., a] + 1) * 0.5 * W - 0.5).astype(np.int64)          # grid_sample, align_corners=False
        iy = np.floor((coords[..., b] + 1) * 0.5 * H - 0.5).astype(np.int64)
        xs = np.stack([ix, ix + 1, ix, ix + 1], -1).reshape(T, -1)
        ys = np.stack([iy, iy, iy + 1, iy + 1], -1).reshape(T, -1)
        key = ys * (W + 4) + xs
        distinct[:, p] = [len(np.unique(k)) for k in key]
        bbox[:, p] = (xs.max(1) - xs.min(1) + 1) * (ys.max(1) - ys.min(1) + 1)
    return distinct, bbox


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "cfg5_render_ws.npz"), allow_pickle=True)
    seed, R, H, D, Ni = (int(z[k]) for k in ("seed", "R", "H", "D", "Ni"))
    rng = np.random.RandomState(seed)
    N = int(z["N"])
    base = rng.randn(N, 96, H, H).astype(np.float32)
    mu = rng.randn(1, 96, 1, 1).astype(np.float32) * 0.7
    sd = np.exp(rng.randn(1, 96, 1, 1).astype(np.float32) * 0.5)
    planes = (base * sd + mu).astype(np.float32)[:1]
    dec = orc.random_decoder(seed + 1, bias_scale=0.3)
    opts = ast.literal_eval(str(z["options"]))
    o, d = orc.ray_sampler(z["cam2world"][:1], z["intrinsics"][:1], R)
    trng = np.random.RandomState(7)
    tiles = [(int(trng.randint(0, R // 4)), int(trng.randint(0, R // 8))) for _ in range(n_tiles)]
    idx = np.array([[(ty * 4 + j // 8) * R + tx * 8 + (j & 7) for j in range(32)] for ty, tx in tiles]).reshape(-1)
    oo, dd = o[:, idx], d[:, idx]
    M = idx.size
    u_c = trng.rand(1, M, D).astype(np.float32)
    u_f = trng.rand(M, Ni).astype(np.float32)
    normed, denormed, _, _ = orc.synthesis_planes(planes)
    out = orc.render(normed, denormed, dec, oo, dd, opts, u_c, u_f, return_taps=True)
    taps = out[-1]
    scale = 2.0 / float(opts["box_warp"])
    for name, depths in (("coarse pass (stratified depths)", taps["depths_coarse"]), ("final pass (merged depth lists)", taps["depths_all"])):
        t = depths.reshape(n_tiles, 32, -1)                                           # [tile, ray, sample]
        S = t.shape[2]
        dist, box = [], []
        for k in range(S):
            c = (oo[0].reshape(n_tiles, 32, 3) + t[:, :, k:k + 1] * dd[0].reshape(n_tiles, 32, 3)) * scale
            a, b = footprint(c, H, H)
            dist.append(a); box.append(b)
        dist, box = np.stack(dist), np.stack(box)                                     # [S, tile, plane]
        spread = (t.max(1) - t.min(1))                                                # depth spread of a tile at one sample index
        print(f"{name}: {n_tiles} tiles x {S} steps; depth spread inside a tile at one index: median {np.median(spread):.4f}, 95 % {np.percentile(spread, 95):.4f} "
              f"(one stratum = {(opts['ray_end'] - opts['ray_start']) / (D - 1):.4f})")
        for p, pn in enumerate(("p0 (x, y)", "p1 (x, z)", "p2 (z, x)")):
            print(f"   {pn}: distinct texels of 128 taps: median {np.median(dist[..., p]):.0f}, mean {dist[..., p].mean():.1f}, 95 % {np.percentile(dist[..., p], 95):.0f}; "
                  f"bounding box (texels): median {np.median(box[..., p]):.0f}, mean {box[..., p].mean():.1f}, 95 % {np.percentile(box[..., p], 95):.0f}")


if __name__ == "__main__":
    main()
