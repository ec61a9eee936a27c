"""Golden vectors for the renderer BACKWARD, produced by the REFERENCE under torch autograd on CPU.

Runs only in the build container (needs /root/reference).  For each case: seeded plane sets (norm, denorm) as leaves
with requires_grad, the reference's DisentangledImportanceRenderer.forward (renderer.py:301-363) with injected jitter,
random cotangents for its four outputs, loss = sum(out * cotangent), loss.backward() -> d loss / d norm_planes and
d loss / d denorm_planes.  The analytic restatement (oracle/render_backward_oracle.py) is checked against these
gradients before anything is written.

    python oracle/gen_golden_backward.py        # writes tests/golden/backward_*.npz
"""
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from training.triplane import TriPlaneGenerator  # noqa: E402  (reference)
from training.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from training.volumetric_rendering.renderer import DisentangledImportanceRenderer  # noqa: E402

from oracle import render_oracle as orc  # noqa: E402
from oracle import render_backward_oracle as bwd  # noqa: E402
from oracle.gen_golden import InjectRand, cams, ref_decoder, smooth_planes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(True)          # oracle.gen_golden switches autograd off at import


def gen_case(tag, seed, N, R, H, D, Ni, angles, white_back=False, swap=False, box_warp=1.0, sigma_bias=0.0):
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    dec_np = orc.random_decoder(seed + 1, bias_scale=0.3)
    if sigma_bias:                                   # denser volumes: transmittance really falls along the ray
        dec_np["geo_net.2.bias"] = dec_np["geo_net.2.bias"].copy()
        dec_np["geo_net.2.bias"][0] += np.float32(sigma_bias)
    c2w, K = cams(angles)
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, max(Ni, 1)).astype(np.float32)[:, :Ni]
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=2.25, ray_end=3.3, box_warp=box_warp,
                disparity_space_sampling=False, clamp_mode="softplus", white_back=white_back)
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)
    with torch.no_grad():
        tp = torch.from_numpy(planes)
        norm, mean, std = TriPlaneGenerator.normalize_plane(G, tp)
        denorm = tp
        if swap:
            denorm = TriPlaneGenerator.denormalize_plane(G, norm, mean.flip(0), std.flip(0))
        o, d = RaySampler()(c2w, K, R)
    norm5 = norm.reshape(N, 3, 32, H, H).clone().requires_grad_(True)
    den5 = denorm.reshape(N, 3, 32, H, H).clone().requires_grad_(True)
    rend = DisentangledImportanceRenderer()
    taps = {}
    orig_unify = rend.unify_samples

    def unify(d1, *rest):
        res = orig_unify(d1, *rest)
        taps["depths_all"] = res[0].detach().numpy().copy()
        return res

    rend.unify_samples = unify
    dec = ref_decoder(dec_np).requires_grad_(False)
    q = [u_c] + ([u_f] if Ni > 0 else [])
    with InjectRand(q):
        rgb, seg, depth, wsum = rend(norm5, den5, dec, o, d, opts)
    cot = dict(rgb=rng.randn(N, M, 32).astype(np.float32), seg=rng.randn(N, M, 15).astype(np.float32),
               depth=rng.randn(N, M, 1).astype(np.float32), wsum=rng.randn(N, M, 1).astype(np.float32))
    loss = sum((v * torch.from_numpy(cot[k])).sum() for k, v in (("rgb", rgb), ("seg", seg), ("depth", depth), ("wsum", wsum)))
    loss.backward()
    g_norm, g_den = norm5.grad.numpy(), den5.grad.numpy()
    if Ni > 0:
        depths_all = taps["depths_all"].reshape(N, M, D + Ni)
    else:
        depths_all = orc.sample_stratified(N, M, opts["ray_start"], opts["ray_end"], D, u_c)
    # the analytic restatement against the reference's autograd
    on, od = bwd.render_backward(norm5.detach().numpy(), den5.detach().numpy(), dec_np, o.numpy(), d.numpy(), depths_all,
                                 opts, cot["rgb"], cot["seg"], cot["depth"], cot["wsum"])
    for name, mine, ref in (("grad_norm", on, g_norm), ("grad_denorm", od, g_den)):
        scale = float(np.abs(ref).max())
        err = float(np.abs(mine - ref).max())
        print(f"    oracle vs reference autograd  {name:12s} max-abs {err:.3e}  (max |grad| {scale:.3e})")
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)
    print(f"    wsum range {float(wsum.min()):.3f} .. {float(wsum.max()):.3f}")
    opts_s = {k: (v if not isinstance(v, bool) else int(v)) for k, v in opts.items()}
    np.savez_compressed(
        os.path.join(OUT, f"backward_{tag}.npz"),
        norm_planes=norm5.detach().numpy(), denorm_planes=den5.detach().numpy(), cam2world=c2w.numpy(), intrinsics=K.numpy(),
        origins=o.numpy(), dirs=d.numpy(), R=R, u_coarse=u_c, u_fine=u_f, options=np.array(repr(opts_s)),
        depths_all=depths_all.astype(np.float32),
        **{"dec." + k: v for k, v in dec_np.items()},
        **{"cot." + k: v for k, v in cot.items()},
        **{"out.rgb": rgb.detach().numpy(), "out.seg": seg.detach().numpy(), "out.depth": depth.detach().numpy(),
           "out.wsum": wsum.detach().numpy()},
        grad_norm=g_norm.astype(np.float32), grad_denorm=g_den.astype(np.float32),
        torch_version=np.array(torch.__version__))
    print(f"  wrote backward_{tag}.npz")


class InjectRandn:
    """Patch torch.randn_like so the reference's density noise (renderer.py:285-286) consumes our normals, in call order."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __enter__(self):
        self._rn = torch.randn_like

        def randn_like(x, *a, **k):
            u = self.queue.pop(0)
            assert u.size == x.numel(), (u.shape, x.shape)
            return torch.from_numpy(u).reshape(x.shape).to(x.dtype)

        torch.randn_like = randn_like
        return self

    def __exit__(self, *a):
        torch.randn_like = self._rn
        assert not self.queue, "reference drew fewer normal tensors than injected"


def gen_variant(tag, seed, N, R, H, D, Ni, angles, density_noise=0.0, segosg=False, sigma_bias=2.0):
    """Round 6: the plane gradients of the two ablation paths the reference gets from autograd and round 5 still raised on -
    `density_noise` (renderer.py:285-286: randn_like * density_noise added to sigma in BOTH run_model calls of a two-pass render;
    the normals are injected so that the HIP path can be given the same ones through nfe_render_args.density_noise_values) and
    SegmentationOSGDecoder (triplane.py:192-230, `disable_alignment`: one plane tensor feeds both arguments and is the leaf)."""
    from training.triplane import SegmentationOSGDecoder
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    if segosg:
        dec_np = orc.random_segmentation_decoder(seed + 1, bias_scale=0.3)
        dec_np["net.2.bias"] = dec_np["net.2.bias"].copy(); dec_np["net.2.bias"][0] += np.float32(sigma_bias)
        dec = SegmentationOSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32, "decoder_seg_dim": 15})
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in dec_np.items()})
        dec = dec.eval().requires_grad_(False)
    else:
        dec_np = orc.random_decoder(seed + 1, bias_scale=0.3)
        dec_np["geo_net.2.bias"] = dec_np["geo_net.2.bias"].copy(); dec_np["geo_net.2.bias"][0] += np.float32(sigma_bias)
        dec = ref_decoder(dec_np).requires_grad_(False)
    c2w, K = cams(angles)
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Ni).astype(np.float32)
    nz_c = rng.randn(N, M, D).astype(np.float32)
    nz_f = rng.randn(N, M, Ni).astype(np.float32)            # in the order sample_importance returns the fine samples (u_fine's)
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                disparity_space_sampling=False, clamp_mode="softplus", white_back=False)
    if density_noise:
        opts["density_noise"] = float(density_noise)
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)
    with torch.no_grad():
        tp = torch.from_numpy(planes)
        norm, mean, std = TriPlaneGenerator.normalize_plane(G, tp)
        o, d = RaySampler()(c2w, K, R)
    if segosg:                                                # triplane.py:119 with disable_disentangle: the raw planes, twice
        leaf = tp.reshape(N, 3, 32, H, H).clone().requires_grad_(True)
        norm5 = den5 = leaf
    else:
        norm5 = norm.reshape(N, 3, 32, H, H).clone().requires_grad_(True)
        den5 = TriPlaneGenerator.denormalize_plane(G, norm, mean.flip(0), std.flip(0)).reshape(N, 3, 32, H, H).clone().requires_grad_(True)
    rend = DisentangledImportanceRenderer()
    taps = {}
    orig_unify, orig_imp, orig_strat = rend.unify_samples, rend.sample_importance, rend.sample_stratified

    def unify(d1, c1, s1, dens1, d2, c2, s2, dens2):
        taps["d_coarse"], taps["d_fine"] = d1.detach().numpy().copy(), d2.detach().numpy().copy()
        res = orig_unify(d1, c1, s1, dens1, d2, c2, s2, dens2)
        taps["depths_all"] = res[0].detach().numpy().copy()
        return res

    rend.unify_samples = unify
    rq = [nz_c.reshape(N, M * D, 1), nz_f.reshape(N, M * Ni, 1)] if density_noise else []
    with InjectRand([u_c, u_f]), InjectRandn(rq):
        rgb, seg, depth, wsum = rend(norm5, den5, dec, o, d, opts)
    cot = dict(rgb=rng.randn(N, M, 32).astype(np.float32), seg=rng.randn(N, M, 15).astype(np.float32),
               depth=rng.randn(N, M, 1).astype(np.float32), wsum=rng.randn(N, M, 1).astype(np.float32))
    loss = sum((v * torch.from_numpy(cot[k])).sum() for k, v in (("rgb", rgb), ("seg", seg), ("depth", depth), ("wsum", wsum)))
    loss.backward()
    g_norm = np.zeros_like(planes.reshape(N, 3, 32, H, H)) if segosg else norm5.grad.numpy()
    g_den = (leaf.grad if segosg else den5.grad).numpy()
    depths_all = taps["depths_all"].reshape(N, M, D + Ni)
    d_fine = taps["d_fine"].reshape(N, M, Ni)
    # the normals in the HIP library's draw order: coarse sample k at k, the fine sample of ascending rank r at D + r
    order_f = np.argsort(d_fine, axis=-1, kind="stable")
    noise_values = np.concatenate([nz_c, np.take_along_axis(nz_f, order_f, -1)], -1).astype(np.float32)
    # ... and per MERGED sample, for the analytic restatement
    d_cat = np.concatenate([taps["d_coarse"].reshape(N, M, D), d_fine], -1)
    order_all = np.argsort(d_cat, axis=-1, kind="stable")
    assert np.array_equal(np.take_along_axis(d_cat, order_all, -1), depths_all)
    sig_off = np.take_along_axis(np.concatenate([nz_c, nz_f], -1), order_all, -1) * np.float32(density_noise) if density_noise else None
    on, od = bwd.render_backward(norm5.detach().numpy(), den5.detach().numpy(), dec_np, o.numpy(), d.numpy(), depths_all,
                                 opts, cot["rgb"], cot["seg"], cot["depth"], cot["wsum"], sigma_offset=sig_off)
    for name, mine, ref in (("grad_norm", on, g_norm), ("grad_denorm", od, g_den)):
        scale = float(np.abs(ref).max()) if np.abs(ref).max() > 0 else 1.0
        err = float(np.abs(mine - ref).max())
        print(f"    oracle vs reference autograd  {name:12s} max-abs {err:.3e}  (max |grad| {scale:.3e})")
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)
    print(f"    wsum range {float(wsum.min()):.3f} .. {float(wsum.max()):.3f}")
    opts_s = {k: (v if not isinstance(v, bool) else int(v)) for k, v in opts.items()}
    np.savez_compressed(
        os.path.join(OUT, f"backward_{tag}.npz"),
        norm_planes=norm5.detach().numpy(), denorm_planes=den5.detach().numpy(), cam2world=c2w.numpy(), intrinsics=K.numpy(),
        origins=o.numpy(), dirs=d.numpy(), R=R, u_coarse=u_c, u_fine=u_f, options=np.array(repr(opts_s)),
        depths_all=depths_all.astype(np.float32), noise_values=noise_values,
        sigma_offset=(sig_off if sig_off is not None else np.zeros((N, M, D + Ni))).astype(np.float32),
        **{"dec." + k: v for k, v in dec_np.items()},
        **{"cot." + k: v for k, v in cot.items()},
        **{"out.rgb": rgb.detach().numpy(), "out.seg": seg.detach().numpy(), "out.depth": depth.detach().numpy(),
           "out.wsum": wsum.detach().numpy()},
        grad_norm=g_norm.astype(np.float32), grad_denorm=g_den.astype(np.float32),
        torch_version=np.array(torch.__version__))
    print(f"  wrote backward_{tag}.npz")


def gen_full_size(name="fullsize_backward", seed=911, N=1, R=128, H=256, D=48, Ni=48, n_keep=40000):
    """The editing configuration at its real size (train.py:306-307: 128^2 rays, 48 + 48 samples, 256^2 planes, two plane sets
    with different statistics) through the reference renderer under autograd.  Planes, decoder, jitter and cotangents are
    regenerated from the seed by the test; the fixture keeps `n_keep` randomly chosen entries of each gradient and fp64
    per-(plane, channel) sums of all of them."""
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    dec_np = orc.random_decoder(seed + 1, bias_scale=0.3)
    dec_np["geo_net.2.bias"] = dec_np["geo_net.2.bias"].copy()
    dec_np["geo_net.2.bias"][0] += np.float32(2.0)
    c2w, K = cams([(0.3, -0.15)])
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Ni).astype(np.float32)
    cot = dict(rgb=rng.randn(N, M, 32).astype(np.float32), seg=rng.randn(N, M, 15).astype(np.float32),
               depth=rng.randn(N, M, 1).astype(np.float32), wsum=rng.randn(N, M, 1).astype(np.float32))
    new_mu = rng.randn(1, 96, 1, 1).astype(np.float32) * 0.5
    new_sd = np.exp(rng.randn(1, 96, 1, 1).astype(np.float32) * 0.3)
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                disparity_space_sampling=False, clamp_mode="softplus", white_back=False)
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)
    with torch.no_grad():
        norm, mean, std = TriPlaneGenerator.normalize_plane(G, torch.from_numpy(planes))
        denorm = TriPlaneGenerator.denormalize_plane(G, norm, torch.from_numpy(new_mu), torch.from_numpy(new_sd))
        o, d = RaySampler()(c2w, K, R)
    norm5 = norm.reshape(N, 3, 32, H, H).clone().requires_grad_(True)
    den5 = denorm.reshape(N, 3, 32, H, H).clone().requires_grad_(True)
    rend = DisentangledImportanceRenderer()
    dec = ref_decoder(dec_np).requires_grad_(False)
    with InjectRand([u_c, u_f]):
        outs = rend(norm5, den5, dec, o, d, opts)
    loss = sum((v * torch.from_numpy(cot[k])).sum() for k, v in zip(("rgb", "seg", "depth", "wsum"), outs))
    loss.backward()
    g_norm, g_den = norm5.grad.numpy(), den5.grad.numpy()
    idx = np.random.RandomState(seed + 7).choice(g_norm.size, n_keep, replace=False)
    print(f"    max |grad_norm| {np.abs(g_norm).max():.3e}  max |grad_denorm| {np.abs(g_den).max():.3e}  wsum {float(outs[3].mean()):.3f}")
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), seed=seed, N=N, R=R, H=H, D=D, Ni=Ni, cam2world=c2w.numpy(), intrinsics=K.numpy(),
        options=np.array(repr(opts)), idx=idx.astype(np.int64),
        grad_norm=g_norm.reshape(-1)[idx], grad_denorm=g_den.reshape(-1)[idx],
        grad_norm_max=float(np.abs(g_norm).max()), grad_denorm_max=float(np.abs(g_den).max()),
        grad_norm_sum=g_norm.astype(np.float64).sum(axis=(0, 3, 4)), grad_denorm_sum=g_den.astype(np.float64).sum(axis=(0, 3, 4)),
        grad_norm_abs=np.abs(g_norm).astype(np.float64).sum(axis=(0, 3, 4)), grad_denorm_abs=np.abs(g_den).astype(np.float64).sum(axis=(0, 3, 4)),
        rgb_mean=outs[0].detach().numpy().astype(np.float64).mean(axis=(0, 1)), torch_version=np.array(torch.__version__))
    print(f"  wrote {name}.npz")


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_case("single", 901, N=2, R=8, H=16, D=12, Ni=0, angles=[(0.3, -0.2), (-0.4, 0.1)])
    gen_case("two_swap_white", 902, N=2, R=8, H=16, D=10, Ni=10, angles=[(0.2, 0.1), (-0.3, -0.2)], white_back=True, swap=True,
             sigma_bias=3.0)
    gen_case("oob_dense", 903, N=1, R=8, H=16, D=16, Ni=0, angles=[(0.5, 0.3)], box_warp=0.45, sigma_bias=6.0)
    gen_variant("noise", 904, N=2, R=8, H=16, D=10, Ni=10, angles=[(0.25, 0.1), (-0.3, -0.15)], density_noise=0.7)
    gen_variant("segosg", 905, N=2, R=8, H=16, D=10, Ni=10, angles=[(0.2, -0.1), (-0.25, 0.2)], segosg=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "variants":          # only the two round-6 fixtures
        os.makedirs(OUT, exist_ok=True)
        gen_variant("noise", 904, N=2, R=8, H=16, D=10, Ni=10, angles=[(0.25, 0.1), (-0.3, -0.15)], density_noise=0.7)
        gen_variant("segosg", 905, N=2, R=8, H=16, D=10, Ni=10, angles=[(0.2, -0.1), (-0.25, 0.2)], segosg=True)
    elif len(sys.argv) > 1 and sys.argv[1] == "full_size":
        os.makedirs(OUT, exist_ok=True)
        gen_full_size()
    else:
        main()
        gen_full_size()
