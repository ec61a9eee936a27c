"""GPU edge cases of the render path against the oracle: ragged ray counts, broadcast planes, rectangular
planes, minimum / maximum sample counts, rays that miss the volume, empty point queries."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import render_oracle as orc
from tests._golden import max_abs

pytestmark = pytest.mark.gpu
TOL = 1e-3
NAMES = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
         "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def run_both(dev, planes_norm, planes_den, dec, o, d, opts, u_c, u_f=None):
    """HIP (dual-plane entry, explicit rays) vs the C oracle on identical inputs."""
    from nerffaceediting_amd import ops
    want = c_oracle.render(planes_norm, planes_den, dec, o, d, opts, u_c, u_f)
    decp = ops.decoder_pack(*[t(dec[k], dev) for k in NAMES])
    pg, pa = ops.plane_pack(t(planes_norm, dev)), ops.plane_pack(t(planes_den, dev))
    got = ops.render(pg, pa, decp, opts, origins=t(o, dev), dirs=t(d, dev), u_coarse=t(u_c, dev),
                     u_fine=None if u_f is None else t(u_f, dev))
    return [g.cpu().numpy() for g in got], want


def rays(rng, N, M, miss=False):
    o = np.tile(np.array([0.0, 0.0, 2.7], np.float32), (N, M, 1)) + rng.randn(N, M, 3).astype(np.float32) * 0.02
    tgt = rng.uniform(-0.45, 0.45, (N, M, 3)).astype(np.float32) + (5.0 if miss else 0.0)
    d = tgt - o
    return o, (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("N,M,Np,H,W,D,Di", [
    (1, 25, 1, 16, 16, 8, 0),        # ragged: 25 rays in one 32-ray block
    (3, 70, 1, 16, 16, 6, 6),        # three views sharing ONE plane set (broadcast), ragged blocks, two-pass
    (2, 33, 2, 12, 20, 5, 4),        # rectangular planes (H != W), minimum two-pass D
    (1, 40, 1, 8, 8, 2, 0),          # minimum depth_resolution
    (1, 32, 1, 8, 8, 256, 0),        # maximum depth_resolution
    (1, 32, 1, 8, 8, 130, 126),      # long two-pass lists
])
def test_shapes(N, M, Np, H, W, D, Di, dev):
    rng = np.random.RandomState(N * 1000 + M)
    pn = rng.randn(Np, 3, 32, H, W).astype(np.float32)
    pd = (rng.randn(Np, 3, 32, H, W) * 0.7 + 0.2).astype(np.float32)
    dec = orc.random_decoder(M, bias_scale=0.2)
    o, d = rays(rng, N, M)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1)
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Di).astype(np.float32) if Di else None
    got, want = run_both(dev, pn, pd, dec, o, d, opts, u_c, u_f)
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), got, want):
        assert g.shape == w.shape
        assert max_abs(g, w) <= TOL, k


def test_all_rays_miss_the_volume(dev):
    """Every sample falls outside the planes: features are zero, weights come from the biases only; the depth
    clamp and nan_to_num (ray_marcher.py:93-94) must give finite numbers equal to the oracle's."""
    rng = np.random.RandomState(5)
    N, M, D = 1, 48, 10
    pn = rng.randn(1, 3, 32, 8, 8).astype(np.float32)
    dec = orc.random_decoder(3, bias_scale=0.0)
    dec["geo_net.2.bias"][0] = -40.0                      # sigma -> softplus(-41) ~ 1e-18: weights underflow to 0
    o, d = rays(rng, N, M, miss=True)
    opts = dict(depth_resolution=D, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1)
    u_c = rng.rand(N, M, D).astype(np.float32)
    got, want = run_both(dev, pn, pn, dec, o, d, opts, u_c)
    assert np.isfinite(got[2]).all()
    assert max_abs(got[2], want[2]) <= TOL and max_abs(got[3], want[3]) <= 1e-6
    assert max_abs(got[0], want[0]) <= TOL


def test_empty_point_query_and_bad_arguments(dev):
    from nerffaceediting_amd import ops
    rng = np.random.RandomState(1)
    p = ops.plane_pack(t(rng.randn(1, 96, 8, 8), dev))
    dec = orc.random_decoder(1)
    decp = ops.decoder_pack(*[t(dec[k], dev) for k in NAMES])
    out = ops.point_query(p, p, decp, torch.zeros(1, 0, 3, device=dev), 1.0)
    assert out["rgb"].shape == (1, 0, 32) and out["sigma"].shape == (1, 0, 1)
    o, d = rays(rng, 1, 8)
    with pytest.raises(RuntimeError, match="importance sampling needs depth_resolution >= 4"):
        ops.render(p, p, decp, dict(depth_resolution=3, depth_resolution_importance=2, ray_start=2.25, ray_end=3.3, box_warp=1),
                   origins=t(o, dev), dirs=t(d, dev))
    with pytest.raises(RuntimeError, match="out of"):
        ops.render(p, p, decp, dict(depth_resolution=257, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1),
                   origins=t(o, dev), dirs=t(d, dev))
    with pytest.raises(AssertionError):
        ops.render(p, p, decp, dict(depth_resolution=8, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1,
                                    clamp_mode="relu"), origins=t(o, dev), dirs=t(d, dev))


@pytest.mark.parametrize("Di", [0, 10])
def test_density_noise_matches_oracle(Di, dev):
    """rendering_options['density_noise'] (renderer.py:285-286): sigma += N(0,1) * std.  The normals are Philox draws keyed by
    (seed, ray, sample depth) - the numpy oracle restates the same draw - so the noisy render is reproducible, agrees with the
    oracle, changes with the seed, and a coarse sample keeps its noise when it is re-evaluated in the final pass."""
    from nerffaceediting_amd import ops
    N, M, H, D = 2, 70, 16, 12
    rng = np.random.RandomState(123 + Di)
    pn = rng.randn(N, 3, 32, H, H).astype(np.float32)
    pd = (rng.randn(N, 3, 32, H, H) * 0.7 + 0.2).astype(np.float32)
    dec = orc.random_decoder(5, bias_scale=0.2)
    o, d = rays(rng, N, M)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1, density_noise=0.7)
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Di).astype(np.float32) if Di else None
    seed = 987654321012
    want = orc.render(pn, pd, dec, o, d, opts, u_c, u_f, noise_seed=seed)
    quiet = orc.render(pn, pd, dec, o, d, dict(opts, density_noise=0), u_c, u_f)
    decp = ops.decoder_pack(*[t(dec[k], dev) for k in NAMES])
    pg, pa = ops.plane_pack(t(pn, dev)), ops.plane_pack(t(pd, dev))
    kw = dict(origins=t(o, dev), dirs=t(d, dev), u_coarse=t(u_c, dev), u_fine=None if u_f is None else t(u_f, dev))
    got = [g.cpu().numpy() for g in ops.render(pg, pa, decp, opts, seed=seed, **kw)]
    again = [g.cpu().numpy() for g in ops.render(pg, pa, decp, opts, seed=seed, **kw)]
    other = [g.cpu().numpy() for g in ops.render(pg, pa, decp, opts, seed=seed + 1, **kw)]
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), got, want):
        assert max_abs(g, w) <= (2e-3 if k == "depth" else 2e-4), k
    assert all(np.array_equal(a, b) for a, b in zip(got, again))
    assert max_abs(got[0], other[0]) > 1e-3 and max_abs(got[0], quiet[0]) > 1e-3
    with pytest.raises(RuntimeError):
        ops.render(pg, pa, decp, dict(opts, density_noise=-1.0), seed=seed, **kw)


def test_point_query_density_noise_matches_oracle(dev):
    """run_model with rendering_options['density_noise'] (renderer.py:285-286) on caller-supplied points: the normal of
    point (n, m) is the Philox draw keyed by (seed; n*P + m, 0), which the oracle restates."""
    from nerffaceediting_amd import ops
    rng = np.random.RandomState(12)
    N, H, P = 2, 16, 777
    planes = (rng.randn(N, 96, H, H) * 1.1).astype(np.float32)
    dec = orc.random_decoder(13, bias_scale=0.2)
    coords = (rng.rand(N, P, 3).astype(np.float32) - 0.5) * 1.1
    seed = 0xABCDEF0123
    opts = dict(box_warp=1, density_noise=0.6)
    want = orc.run_model(*orc.synthesis_planes(planes)[:2], dec, coords, opts, noise_seed=seed)
    quiet = orc.run_model(*orc.synthesis_planes(planes)[:2], dec, coords, dict(opts, density_noise=0))
    assert float(np.abs(want[1] - quiet[1]).std()) > 0.2                    # the noise is really there
    p = t(planes, dev)
    mean, std = ops.plane_stats(p)
    packed = ops.plane_pack(p)
    decp = ops.decoder_pack(*[t(dec[k], dev) for k in NAMES])
    got = ops.point_query(packed, packed, decp, t(coords, dev), 1.0, affines=ops.make_affine(mean, std), density_noise=0.6, seed=seed)
    assert float(np.abs(got["sigma"].cpu().numpy() - want[1]).max()) <= 1e-3
    assert float(np.abs(got["rgb"].cpu().numpy() - want[0]).max()) <= 1e-3
    # through the module interface (seed drawn from torch's generator: reproducible under manual_seed)
    from nerffaceediting_amd.training.triplane import DisentangledOSGDecoder
    from nerffaceediting_amd.training.volumetric_rendering.renderer import DisentangledImportanceRenderer
    d = DisentangledOSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32, "decoder_seg_dim": 15})
    d.load_state_dict({k: torch.from_numpy(v) for k, v in dec.items()})
    d = d.to(dev)
    norm5, den5 = (t(a, dev) for a in orc.synthesis_planes(planes)[:2])
    rend = DisentangledImportanceRenderer()
    torch.manual_seed(5); a = rend.run_model(norm5, den5, d, t(coords, dev), None, opts)["sigma"]
    torch.manual_seed(5); b = rend.run_model(norm5, den5, d, t(coords, dev), None, opts)["sigma"]
    c = rend.run_model(norm5, den5, d, t(coords, dev), None, opts)["sigma"]
    assert torch.equal(a, b) and not torch.equal(a, c)
    with pytest.raises(RuntimeError, match="density_noise"):
        ops.point_query(packed, packed, decp, t(coords, dev), 1.0, density_noise=-0.5)


@pytest.mark.parametrize("math", ["bf16x3", "fp32"])
def test_large_preactivations_are_not_clamped(math, dev):
    """Hidden pre-activations far above torch's Softplus threshold (x > 20 -> x; edited / optimised planes can drive them there):
    softplus must return x itself, for any x - an earlier log2(1 + 2^min(y,126)) form saturated at x = 87.3.  seg is
    a linear read-out of the hidden layer, so a clamp shows up there at full size (hundreds); bar relative to the magnitude:
    split-bf16 drops the lo x lo term of every product (2^-16 relative), the exact-fp32 mode is held to fp32 rounding."""
    from nerffaceediting_amd import ops
    rng = np.random.RandomState(77)
    N, M, H, D = 1, 64, 16, 8
    pn = (rng.randn(N, 3, 32, H, H) * 400.0).astype(np.float32)         # pre-activations of a few hundred (rms ~ 230)
    dec = orc.random_decoder(9, bias_scale=0.2)
    o, d = rays(rng, N, M)
    coords = (rng.rand(N, 500, 3).astype(np.float32) - 0.5) * 0.9
    want = orc.run_model(pn, pn, dec, coords, dict(box_warp=1))
    rel = 5e-5 if math == "bf16x3" else 1e-5
    hid = np.abs(want[2]).max()
    assert hid > 300.0, hid                                             # seg magnitudes only reachable with hidden units >> 87.3
    decp = ops.decoder_pack(*[t(dec[k], dev) for k in NAMES])
    p = ops.plane_pack(t(pn, dev))
    got = ops.point_query(p, p, decp, t(coords, dev), 1.0, decoder_math=math)
    for k, i in (("rgb", 0), ("sigma", 1), ("seg", 2)):
        w = want[i]
        e = float(np.abs(got[k].cpu().numpy() - w).max())
        assert e <= 1e-3 + rel * float(np.abs(w).max()), (k, e)
    opts = dict(depth_resolution=D, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1)
    u_c = rng.rand(N, M, D).astype(np.float32)
    wr = c_oracle.render(pn, pn, dec, o, d, opts, u_c, None)
    gr = ops.render(p, p, decp, opts, origins=t(o, dev), dirs=t(d, dev), u_coarse=t(u_c, dev), decoder_math=math)
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), gr, wr):
        e = max_abs(g.cpu().numpy(), w)
        assert e <= 1e-3 + rel * float(np.abs(w).max()), (k, e)


@pytest.mark.parametrize("Di", [0, 6])
def test_infinite_density_does_not_poison_the_ray(Di, dev):
    """sigma = +inf at every sample (geometry output bias +inf): the reference's first interval gets alpha = 1 and takes all the
    weight (ray_marcher.py:76-88).  The kernel composites the first sample of a march / depth segment with a zero-length
    interval: that must be alpha = 0 by construction, not 1 - exp(-(inf * 0)) = NaN."""
    rng = np.random.RandomState(31)
    N, M, H, D = 1, 40, 8, 9
    pn = rng.randn(N, 3, 32, H, H).astype(np.float32)
    dec = orc.random_decoder(4, bias_scale=0.1)
    dec["geo_net.2.bias"][0] = np.inf
    o, d = rays(rng, N, M)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1)
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Di).astype(np.float32) if Di else None
    want = orc.render(pn, pn, dec, o, d, opts, u_c, u_f)
    assert all(np.isfinite(w).all() for w in want) and np.allclose(want[3], 1.0, atol=1e-6)
    got, _ = run_both(dev, pn, pn, dec, o, d, opts, u_c, u_f)
    for k, g, w in zip(("rgb", "seg", "depth", "wsum"), got, want):
        assert np.isfinite(g).all(), k
        assert max_abs(g, w) <= TOL, k
