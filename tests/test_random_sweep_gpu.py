"""GPU: seeded random sweep of the render core and its backward over shapes and options (ragged ray counts, rectangular
planes, broadcast plane sets, white_back, disparity sampling, box_warp, dense / thin volumes) against the C oracle
(forward) and the analytic backward oracle.  Every case is a different corner of the argument space; sizes keep the CPU
side in seconds."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import render_backward_oracle as bwd
from oracle import render_oracle as orc
from tests._golden import max_abs

pytestmark = pytest.mark.gpu
NAMES = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
         "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def draw_case(seed, backward=False):
    rng = np.random.RandomState(10_000 + seed)
    N = int(rng.choice([1, 2, 3]))
    M = int(rng.choice([7, 31, 32, 33, 64, 100])) if not backward else int(rng.choice([9, 36, 64, 70]))
    Np = 1 if (N > 1 and rng.rand() < 0.3) else N
    H, W = int(rng.choice([6, 9, 16, 24])), int(rng.choice([6, 11, 16, 24]))
    D = int(rng.choice([4, 5, 9, 16, 33])) if not backward else int(rng.choice([4, 7, 12]))
    Di = int(rng.choice([0, 0, 4, 7, 16, 40])) if not backward else int(rng.choice([0, 5, 9]))
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=float(rng.uniform(1.9, 2.4)),
                ray_end=float(rng.uniform(3.0, 3.6)), box_warp=float(rng.choice([0.6, 1.0, 1.0, 1.7])),
                white_back=bool(rng.rand() < 0.3), disparity_space_sampling=bool(rng.rand() < 0.2 and not backward),
                clamp_mode="softplus")
    pn = (rng.randn(Np, 3, 32, H, W) * rng.uniform(0.5, 1.5)).astype(np.float32)
    same = rng.rand() < 0.25
    pd = pn if same else (rng.randn(Np, 3, 32, H, W) * 0.8 + 0.1).astype(np.float32)
    dec = orc.random_decoder(seed + 77, bias_scale=float(rng.uniform(0.0, 0.5)))
    dec["geo_net.2.bias"][0] += np.float32(rng.choice([-3.0, 0.0, 2.0, 6.0]))       # empty ... opaque volumes
    o = np.tile(np.array([0.0, 0.0, 2.7], np.float32), (N, M, 1)) + rng.randn(N, M, 3).astype(np.float32) * 0.03
    tgt = rng.uniform(-0.55, 0.55, (N, M, 3)).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Di).astype(np.float32) if Di else None
    return dict(N=N, M=M, Np=Np, opts=opts, pn=pn, pd=pd, same=same, dec=dec, o=o, d=d, u_c=u_c, u_f=u_f, rng=rng)


@pytest.mark.parametrize("seed", range(28))
def test_forward_sweep(seed, dev):
    from nerffaceediting_amd import ops
    c = draw_case(seed)
    rep = (lambda a: np.repeat(a, c["N"], 0)) if c["Np"] != c["N"] else (lambda a: a)
    want = c_oracle.render(rep(c["pn"]), rep(c["pd"]), c["dec"], c["o"], c["d"], c["opts"], c["u_c"], c["u_f"])
    decp = ops.decoder_pack(*[t(c["dec"][k], dev) for k in NAMES])
    pg = ops.plane_pack(t(c["pn"], dev))
    pa = pg if c["same"] else ops.plane_pack(t(c["pd"], dev))
    math = "fp32" if seed % 3 == 0 else None
    got = ops.render(pg, pa, decp, c["opts"], origins=t(c["o"], dev), dirs=t(c["d"], dev), u_coarse=t(c["u_c"], dev),
                     u_fine=None if c["u_f"] is None else t(c["u_f"], dev), decoder_math=math)
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), got, want):
        assert max_abs(g.cpu().numpy(), w) <= 1e-3, (seed, k, {kk: vv for kk, vv in c["opts"].items()}, c["N"], c["M"])


@pytest.mark.parametrize("seed", range(10))
def test_backward_sweep(seed, dev):
    from nerffaceediting_amd import ops
    c = draw_case(seed, backward=True)
    N, M, rng = c["N"], c["M"], c["rng"]
    rep = (lambda a: np.repeat(a, N, 0)) if c["Np"] != N else (lambda a: a)
    decp_heads = [t(c["dec"][k], dev) for k in NAMES]
    pg = ops.plane_pack(t(c["pn"], dev))
    pa = pg if c["same"] else ops.plane_pack(t(c["pd"], dev))
    kw = dict(origins=t(c["o"], dev), dirs=t(c["d"], dev))
    out = ops.render(pg, pa, ops.decoder_pack(*decp_heads), c["opts"], u_coarse=t(c["u_c"], dev),
                     u_fine=None if c["u_f"] is None else t(c["u_f"], dev), taps=True, decoder_math="fp32", **kw)
    depths = out[4]["depths_all"]
    S = depths.shape[-1]
    cot = [rng.randn(N, M, 32).astype(np.float32), rng.randn(N, M, 15).astype(np.float32),
           rng.randn(N, M, 1).astype(np.float32), rng.randn(N, M, 1).astype(np.float32)]
    drop = seed % 4                                   # some cotangents absent (NULL pointers in the ABI)
    cots = tuple(None if (i == drop and i > 0) else t(x, dev) for i, x in enumerate(cot))
    zero = lambda i: np.zeros_like(cot[i]) if (i == drop and i > 0) else cot[i]
    gn, gd = bwd.render_backward(rep(c["pn"]), rep(c["pd"]), c["dec"], c["o"], c["d"], depths.cpu().numpy().reshape(N, M, S), c["opts"],
                                 zero(0), zero(1), zero(2), zero(3))
    if c["Np"] != N:
        gn, gd = gn.sum(0, keepdims=True), gd.sum(0, keepdims=True)
    gg, ga = ops.render_backward(pg, pa, decp_heads, 1.0, c["opts"], depths, cots, **kw)
    un = lambda g: g.permute(0, 1, 4, 2, 3).contiguous().cpu().numpy()
    if c["same"]:
        want = gn + gd
        assert gg is ga
        assert float(np.abs(un(gg) - want).max()) <= 1e-3 * float(np.abs(want).max()) + 1e-7, seed
    else:
        for got, want in ((gg, gn), (ga, gd)):
            assert float(np.abs(un(got) - want).max()) <= 1e-3 * float(np.abs(want).max()) + 1e-7, seed
    # the same gradients from the decoder outputs a split-bf16 forward keeps (tap_sample_colors, ABI v11) instead of the re-evaluation
    # pass: arbitrary ray counts (untiled ray blocks), broadcast planes, one or two plane sets, absent cotangents
    out2 = ops.render(pg, pa, ops.decoder_pack(*decp_heads), c["opts"], u_coarse=t(c["u_c"], dev),
                      u_fine=None if c["u_f"] is None else t(c["u_f"], dev), taps=True, sample_colors=True, **kw)
    g2, a2 = ops.render_backward(pg, pa, decp_heads, 1.0, c["opts"], out2[4]["depths_all"], cots, sample_colors=out2[4]["sample_colors"],
                                 sample_colors_resolution=out2[4]["sample_colors_resolution"], **kw)
    r2, ra2 = ops.render_backward(pg, pa, decp_heads, 1.0, c["opts"], out2[4]["depths_all"], cots, **kw)
    # (5e-6: the two forms feed the same float atomics, whose order differs from launch to launch - tools/archive/r04_flake_probe.py measured up
    # to 1.02e-6 between two launches of the SAME form on case 6, whose 6 x 6 planes collect ~280 contributions per texel)
    for got, ref in ((g2, r2), (a2, ra2)):
        assert float((got - ref).abs().max()) <= 5e-6 * float(ref.abs().max()) + 1e-9, seed


@pytest.mark.parametrize("D,Di", [(256, 0), (256, 256), (4, 256), (255, 1), (2, 0)])
def test_sample_count_limits(D, Di, dev):
    """The ends of the sample-count ranges the ABI accepts (depth_resolution 2..NFE_MAX_SAMPLES, importance 0..NFE_MAX_SAMPLES:
    the merged march then has up to 512 samples, importance_kernel's key list and LDS layout are at their largest) against the C
    oracle, forward; and the backward of the largest case against the analytic backward oracle."""
    from nerffaceediting_amd import ops
    rng = np.random.RandomState(4242 + D + 7 * Di)
    N, M, H, W = 2, 40, 16, 24
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.2, ray_end=3.3, box_warp=1.0, white_back=False,
                disparity_space_sampling=False, clamp_mode="softplus")
    pn = rng.randn(N, 3, 32, H, W).astype(np.float32)
    pd = (rng.randn(N, 3, 32, H, W) * 0.8 + 0.1).astype(np.float32)
    dec = orc.random_decoder(D + Di, bias_scale=0.2)
    o = np.tile(np.array([0.0, 0.0, 2.7], np.float32), (N, M, 1)) + rng.randn(N, M, 3).astype(np.float32) * 0.03
    d = rng.uniform(-0.5, 0.5, (N, M, 3)).astype(np.float32) - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Di).astype(np.float32) if Di else None
    want = c_oracle.render(pn, pd, dec, o, d, opts, u_c, u_f)
    heads = [t(dec[k], dev) for k in NAMES]
    pg, pa = ops.plane_pack(t(pn, dev)), ops.plane_pack(t(pd, dev))
    kw = dict(origins=t(o, dev), dirs=t(d, dev))
    got = ops.render(pg, pa, ops.decoder_pack(*heads), opts, u_coarse=t(u_c, dev), u_fine=None if u_f is None else t(u_f, dev), taps=True, **kw)
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), got[:4], want):
        assert max_abs(g.cpu().numpy(), w) <= 1e-3, (D, Di, k)
    depths = got[4]["depths_all"]
    assert depths.shape[-1] == D + Di and bool((depths[..., 1:] >= depths[..., :-1]).all())
    if (D, Di) != (256, 256):
        return
    cot = [rng.randn(N, M, c).astype(np.float32) for c in (32, 15, 1, 1)]
    gn, gd = bwd.render_backward(pn, pd, dec, o, d, depths.cpu().numpy().reshape(N, M, D + Di), opts, *cot)
    gg, ga = ops.render_backward(pg, pa, heads, 1.0, opts, depths, tuple(t(x, dev) for x in cot), **kw)
    un = lambda g: g.permute(0, 1, 4, 2, 3).contiguous().cpu().numpy()
    for got_g, want_g in ((gg, gn), (ga, gd)):
        assert float(np.abs(un(got_g) - want_g).max()) <= 1e-3 * float(np.abs(want_g).max()) + 1e-7


@pytest.mark.parametrize("two_pass", [False, True])
def test_pose_sweep_image_rays(two_pass, dev):
    """Image rays generated in-kernel from 12 cameras (frontal to grazing yaw / pitch, near and far radii, narrow and wide
    fields of view; many rays cross plane borders or miss the box) at 64^2 rays, against the C oracle fed with the oracle's
    own RaySampler rays."""
    from nerffaceediting_amd import ops
    rng = np.random.RandomState(99)
    poses = [(0.0, 0.0, 2.7, 18.837), (0.9, 0.0, 2.7, 18.837), (-1.3, 0.2, 2.7, 18.837), (0.3, 0.7, 2.7, 18.837),
             (0.0, -0.8, 2.7, 18.837), (2.6, 0.1, 2.7, 18.837), (0.4, -0.2, 1.2, 30.0), (0.0, 0.0, 4.0, 12.0),
             (1.57, 0.0, 2.2, 25.0), (-0.4, 0.5, 3.3, 18.837), (0.2, -0.1, 2.7, 45.0), (3.14, 0.0, 2.7, 18.837)]
    N, R, H, D, Di = len(poses), 64, 64, 24, (24 if two_pass else 0)
    c2w = np.concatenate([orc.lookat_pose(np.pi / 2 + y, np.pi / 2 + p, [0, 0, 0.2], r).reshape(1, 4, 4) for (y, p, r, f) in poses])
    K = np.stack([orc.fov_to_intrinsics(f) for (y, p, r, f) in poses])
    planes = (rng.randn(1, 96, H, H) * 1.1 + 0.1).astype(np.float32)
    norm, den, _, _ = orc.synthesis_planes(planes)
    dec = orc.random_decoder(5, bias_scale=0.3)
    dec["geo_net.2.bias"][0] += np.float32(2.5)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=0.6, ray_end=4.6, box_warp=1.0, clamp_mode="softplus")
    u_c = rng.rand(N, R * R, D).astype(np.float32)
    u_f = rng.rand(N * R * R, Di).astype(np.float32) if Di else None
    o, d = orc.ray_sampler(c2w, K, R)
    want = c_oracle.render(np.repeat(norm, N, 0), np.repeat(den, N, 0), dec, o, d, opts, u_c, u_f)
    p = t(planes, dev)
    mean, std = ops.plane_stats(p)
    got = ops.render(ops.plane_pack(p), ops.plane_pack(p), ops.decoder_pack(*[t(dec[k], dev) for k in NAMES]), opts, cam2world=t(c2w, dev),
                     intrinsics=t(K, dev), resolution=R, affines=[a.repeat(N, 1) for a in ops.make_affine(mean, std)],
                     u_coarse=t(u_c, dev), u_fine=None if u_f is None else t(u_f, dev))
    pts = o[:, :, None, :] + np.linspace(0.6, 4.6, 8)[None, None, :, None] * d[:, :, None, :]
    outside = (np.abs(pts) > 0.5).any(-1).mean()
    assert 0.3 < outside < 0.99                                                    # most samples are outside the box, some inside
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), got, want):
        assert max_abs(g.cpu().numpy(), w) <= 1e-3, k
