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


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "full_size":
        os.makedirs(OUT, exist_ok=True)
        gen_full_size()
    else:
        main()
        gen_full_size()
