"""Generate golden input/output vectors for the render core by running the REFERENCE on CPU.

Runs only in the build container (needs /root/reference; never on the GPU box).  It imports the
reference's own modules unmodified, feeds them seeded inputs with injected jitter (torch.rand /
rand_like patched to pop our u tensors, in the reference's draw order: renderer.py:190 then :237),
stores inputs + outputs as small .npz fixtures under tests/golden/, and checks the numpy oracle
(oracle/render_oracle.py) against the same outputs before writing anything.

    python oracle/gen_golden.py            # regenerate tests/golden/*.npz
"""
import math
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from camera_utils import FOV_to_intrinsics, LookAtPoseSampler  # noqa: E402  (reference)
from training.triplane import DisentangledOSGDecoder, TriPlaneGenerator  # noqa: E402
from training.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from training.volumetric_rendering.renderer import DisentangledImportanceRenderer  # noqa: E402

from oracle import render_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


class InjectRand:
    """Patch torch.rand_like / torch.rand so the reference consumes our jitter tensors."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __enter__(self):
        self._rl, self._r = torch.rand_like, torch.rand

        def rand_like(x, *a, **k):
            u = self.queue.pop(0)
            assert u.size == x.numel(), (u.shape, x.shape)
            return torch.from_numpy(u).reshape(x.shape).to(x.dtype)

        def rand(*shape, **k):
            u = self.queue.pop(0)
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape
            assert u.size == int(np.prod(shape)), (u.shape, shape)
            return torch.from_numpy(u).reshape(shape)

        torch.rand_like, torch.rand = rand_like, rand
        return self

    def __exit__(self, *a):
        torch.rand_like, torch.rand = self._rl, self._r
        assert not self.queue, "reference drew fewer random tensors than injected"


def ref_decoder(dec_np):
    dec = DisentangledOSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32, "decoder_seg_dim": 15})
    sd = {k: torch.from_numpy(v) for k, v in dec_np.items()}
    dec.load_state_dict(sd)
    return dec.eval()


def cams(angles, radius=2.7, pivot=(0, 0, 0.2), fov=18.837):
    c2w = torch.cat([LookAtPoseSampler.sample(math.pi / 2 + y, math.pi / 2 + p, torch.tensor(pivot), radius=radius)
                     for (y, p) in angles], 0)
    K = FOV_to_intrinsics(fov)[None].repeat(len(angles), 1, 1)
    return c2w, K


def smooth_planes(rng, N, H, scale=1.0):
    """Random planes with per-channel mean/std spread (so normalisation matters)."""
    base = rng.randn(N, 96, H, H).astype(np.float32)
    mu = rng.randn(1, 96, 1, 1).astype(np.float32) * 0.7
    sd = np.exp(rng.randn(1, 96, 1, 1).astype(np.float32) * 0.5)
    return (base * sd + mu).astype(np.float32) * np.float32(scale)


def check(name, a, b, tol):
    err = float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if a.size else 0.0
    print(f"    oracle vs reference  {name:14s} max-abs {err:.3e}")
    assert err <= tol, (name, err, tol)


def gen_render_case(tag, seed, N, R, H, D, Ni, angles, white_back=False, swap=False, box_warp=1.0,
                    ray_start=2.25, ray_end=3.3, disparity=False, bias_scale=0.3, auto=False):
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    dec_np = orc.random_decoder(seed + 1, bias_scale=bias_scale)
    c2w, K = cams(angles)
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, max(Ni, 1)).astype(np.float32)[:, :Ni]
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=ray_start, ray_end=ray_end,
                box_warp=box_warp, disparity_space_sampling=disparity, clamp_mode="softplus", white_back=white_back)
    if auto:
        opts["ray_start"] = opts["ray_end"] = "auto"
    # reference -------------------------------------------------------------------------------
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)       # only the three plane helpers are used
    tp = torch.from_numpy(planes)
    norm, mean, std = TriPlaneGenerator.normalize_plane(G, tp)
    denorm = tp
    if swap:   # appearance swap: statistics of the batch-reversed planes (utils.py:152-158 usage)
        denorm = TriPlaneGenerator.denormalize_plane(G, norm, mean.flip(0), std.flip(0))
    norm5 = norm.view(N, 3, 32, H, H)
    den5 = denorm.reshape(N, 3, 32, H, H)
    o, d = RaySampler()(c2w, K, R)
    rend = DisentangledImportanceRenderer()
    taps = {}
    orig_si = rend.sample_importance

    def si(z, w, n):
        taps["weights_coarse"] = w.numpy().copy()
        t = orig_si(z, w, n)
        taps["depths_fine"] = t.numpy().copy()
        return t

    rend.sample_importance = si
    q = [u_c] + ([u_f] if Ni > 0 else [])
    with InjectRand(q):
        rgb, seg, depth, wsum = rend(norm5, den5, ref_decoder(dec_np), o, d, opts)
    ref = dict(rgb=rgb.numpy(), seg=seg.numpy(), depth=depth.numpy(), wsum=wsum.numpy())
    # oracle ----------------------------------------------------------------------------------
    oo, od = orc.ray_sampler(c2w.numpy(), K.numpy(), R)
    check("origins", oo, o.numpy(), 1e-6)
    check("dirs", od, d.numpy(), 1e-6)
    on, odn, omean, ostd = orc.synthesis_planes(planes, *((mean.flip(0).numpy(), std.flip(0).numpy()) if swap else (None, None)))
    check("norm_planes", on, norm5.numpy(), 2e-5)
    check("denorm_planes", odn, den5.numpy(), 2e-5)
    r = orc.render(on, odn, dec_np, oo, od, opts, u_c, u_f if Ni > 0 else None, return_taps=True)
    for k, v in zip(("rgb", "seg", "depth", "wsum"), r[:4]):
        check(k, v, ref[k], 2e-5)
    if Ni > 0:
        check("weights_coarse", r[4]["weights_coarse"], taps["weights_coarse"], 1e-5)
        check("depths_fine", r[4]["depths_fine"], taps["depths_fine"], 1e-4)
    rc = orc.render_chunked(on, odn, dec_np, oo, od, opts, u_c, u_f if Ni > 0 else None, chunk=max(8, M // 3)) \
        if not auto else r
    for k, v in zip(("rgb", "seg", "depth", "wsum"), rc[:4]):
        check(k + "(chunk)", v, ref[k], 2e-5)
    opts_s = {k: (v if not isinstance(v, bool) else int(v)) for k, v in opts.items()}
    np.savez_compressed(
        os.path.join(OUT, f"render_{tag}.npz"),
        planes=planes, cam2world=c2w.numpy(), intrinsics=K.numpy(), R=R, swap=int(swap),
        u_coarse=u_c, u_fine=u_f, options=np.array(repr(opts_s)),
        **{"dec." + k: v for k, v in dec_np.items()},
        **{"out." + k: v for k, v in ref.items()},
        **{"tap." + k: v for k, v in taps.items()},
        torch_version=np.array(torch.__version__))
    print(f"  wrote render_{tag}.npz")


CHUNK_NOTE = ("reference run in ray chunks: ray_marcher.py:94 clamps depth to the [min, max] of the depths tensor of EACH CALL, i.e. per chunk here "
              "(not per whole view); a weighted mean of sampled depths lies inside any chunk's range, so the clamp can bind only on rays whose "
              "weights sum to 0 (depth NaN -> inf -> clamp); min(wsum) over the stored rays is recorded as wsum_min")


def gen_render_full_size(name="fullsize_render", seed=401, N=1, R=512, H=256, D=64, Ni=0, swap=False, stride=61, chunk=32768,
                         angles=((0.3, -0.2),)):
    """Real-size render cases through the reference renderer, in ray chunks (the only cross-ray term, the depth clamp to
    the call's [min,max] of sampled depths, cannot bind a weighted mean of those depths).  Planes, decoder and jitter are
    regenerated from the seed by the tests; the fixture keeps every `stride`-th ray of the reference outputs.
      fullsize_render: BASELINE config-2 size, 512^2 rays x 64 samples, 256^2 planes, one view;
      cfg5_render:     BASELINE config 5 (projector.py:33-34: 96 + 96 samples), 128^2 rays, two views, swapped statistics;
      ffhq_render:     the FFHQ rendering_kwargs (train.py:306-307: 128^2 rays, 48 + 48 samples), two views with
                       swapped appearance statistics (norm_planes != normalised denorm_planes: the editing path);
      cfg5_render_ws:  config 5 at its THROUGHPUT launch shape: two 512^2 views (16 384 ray blocks), 96 + 96 samples, swapped
                       statistics - the size at which the library takes its wave-specialised two-pass kernels."""
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    dec_np = orc.random_decoder(seed + 1, bias_scale=0.3)
    c2w, K = cams(list(angles))
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, max(Ni, 1)).astype(np.float32)[:, :Ni]
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                disparity_space_sampling=False, clamp_mode="softplus", white_back=False)
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)
    tp = torch.from_numpy(planes)
    norm, mean, std = TriPlaneGenerator.normalize_plane(G, tp)
    denorm = TriPlaneGenerator.denormalize_plane(G, norm, mean.flip(0), std.flip(0)) if swap else tp
    norm5, den5 = norm.view(N, 3, 32, H, H), denorm.reshape(N, 3, 32, H, H)
    o, d = RaySampler()(c2w, K, R)
    rend, dec = DisentangledImportanceRenderer(), ref_decoder(dec_np)
    outs = []
    for a in range(0, M, chunk):
        e = min(M, a + chunk)
        q = [u_c[:, a:e]] + ([u_f.reshape(N, M, Ni)[:, a:e].reshape(-1, Ni)] if Ni > 0 else [])
        with InjectRand(q):
            outs.append([x.numpy() for x in rend(norm5, den5, dec, o[:, a:e], d[:, a:e], opts)])
        print(f"    reference rays {a}..{e}")
    rgb, seg, depth, wsum = (np.concatenate([o_[i] for o_ in outs], 1) for i in range(4))
    idx = np.arange(0, M, stride)
    opts_s = {k: (v if not isinstance(v, bool) else int(v)) for k, v in opts.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, N=N, R=R, H=H, D=D, Ni=Ni, swap=int(swap), stride=stride,
                        cam2world=c2w.numpy(), intrinsics=K.numpy(), options=np.array(repr(opts_s)),
                        rgb=rgb[:, idx], seg=seg[:, idx], depth=depth[:, idx], wsum=wsum[:, idx],
                        rgb_mean=rgb.astype(np.float64).mean(axis=(0, 1)), wsum_mean=float(wsum.astype(np.float64).mean()),
                        wsum_min=float(wsum[:, idx].min()), note=np.array(CHUNK_NOTE), torch_version=np.array(torch.__version__))
    print(f"  wrote {name}.npz")


def gen_legacy_renderer(seed=21, N=2, R=8, H=16, D=12, Ni=12):
    """ImportanceRenderer + OSGDecoder (renderer.py:81-140, triplane.py:167-190): the single-plane-set, single-MLP path."""
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.triplane import OSGDecoder
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H).reshape(N, 3, 32, H, H)
    c2w, K = cams([(0.4, -0.2), (-0.3, 0.1)][:N])
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Ni).astype(np.float32)
    dec = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
    w = {"net.0.weight": rng.randn(64, 32), "net.0.bias": rng.randn(64) * 0.3, "net.2.weight": rng.randn(33, 64), "net.2.bias": rng.randn(33) * 0.3}
    dec.load_state_dict({k: torch.from_numpy(v.astype(np.float32)) for k, v in w.items()})
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                disparity_space_sampling=False, clamp_mode="softplus", white_back=False)
    o, d = RaySampler()(c2w, K, R)
    rend = ImportanceRenderer()
    with InjectRand([u_c, u_f]):
        rgb, depth, wsum = rend(torch.from_numpy(planes), dec.eval(), o, d, opts)
    coords = torch.from_numpy(((rng.rand(N, 200, 3) - 0.5) * 1.1).astype(np.float32))
    pq = rend.run_model(torch.from_numpy(planes), dec, coords, None, opts)
    opts_s = {k: (v if not isinstance(v, bool) else int(v)) for k, v in opts.items()}
    np.savez_compressed(os.path.join(OUT, "legacy_renderer.npz"), planes=planes, cam2world=c2w.numpy(), intrinsics=K.numpy(), R=R,
                        u_coarse=u_c, u_fine=u_f, options=np.array(repr(opts_s)), coords=coords.numpy(),
                        rgb=rgb.numpy(), depth=depth.numpy(), wsum=wsum.numpy(), pq_rgb=pq["rgb"].numpy(), pq_sigma=pq["sigma"].numpy(),
                        **{"dec." + k: v.astype(np.float32) for k, v in w.items()})
    print("  wrote legacy_renderer.npz")


def gen_segmentation_decoder(seed=31, N=2, R=8, H=16, D=12, Ni=12, P=300):
    """The `disable_alignment` ablation: the reference renderer with the reference SegmentationOSGDecoder (triplane.py:192-230),
    both plane arguments = the raw planes (triplane.py:119 with disable_disentangle), two-pass render + run_model."""
    from training.triplane import SegmentationOSGDecoder
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    dec_np = orc.random_segmentation_decoder(seed + 1, bias_scale=0.3)
    dec = SegmentationOSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32, "decoder_seg_dim": 15})
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in dec_np.items()})
    dec.eval()
    c2w, K = cams([(0.25, -0.1), (-0.35, 0.15)])
    M = R * R
    u_c = rng.rand(N, M, D).astype(np.float32)
    u_f = rng.rand(N * M, Ni).astype(np.float32)
    opts = dict(depth_resolution=D, depth_resolution_importance=Ni, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                disparity_space_sampling=False, clamp_mode="softplus", white_back=False)
    p5 = torch.from_numpy(planes).view(N, 3, 32, H, H)
    o, d = RaySampler()(c2w, K, R)
    rend = DisentangledImportanceRenderer()
    with InjectRand([u_c, u_f]):
        rgb, seg, depth, wsum = rend(p5, p5, dec, o, d, opts)
    coords = ((rng.rand(N, P, 3) - 0.5) * 1.1).astype(np.float32)
    pq = rend.run_model(p5, p5, dec, torch.from_numpy(coords), None, opts)
    ref = dict(rgb=rgb.numpy(), seg=seg.numpy(), depth=depth.numpy(), wsum=wsum.numpy())
    r = orc.render(p5.numpy(), p5.numpy(), dec_np, o.numpy(), d.numpy(), opts, u_c, u_f)
    for k, v in zip(("rgb", "seg", "depth", "wsum"), r[:4]):
        check(k, v, ref[k], 2e-5)
    q = orc.run_model(p5.numpy(), p5.numpy(), dec_np, coords, opts)
    for k, v in zip(("rgb", "sigma", "seg"), q):
        check("pq." + k, v, pq[k].numpy(), 2e-5)
    np.savez_compressed(
        os.path.join(OUT, "segdecoder_render.npz"),
        planes=planes, cam2world=c2w.numpy(), intrinsics=K.numpy(), R=R, u_coarse=u_c, u_fine=u_f, options=np.array(repr(opts)),
        coords=coords,
        **{"dec." + k: v for k, v in dec_np.items()},
        **{"out." + k: v for k, v in ref.items()},
        **{"pq." + k: pq[k].numpy() for k in ("rgb", "sigma", "seg")},
        torch_version=np.array(torch.__version__))
    print("  wrote segdecoder_render.npz")


def gen_decoder_forward(seed=41, N=2, M=70):
    """The three decoder modules called directly on sampled features [N,3,M,32] (triplane.py:178-190, 209-230, 249-270)."""
    from training.triplane import OSGDecoder, SegmentationOSGDecoder
    rng = np.random.RandomState(seed)
    fn = (rng.randn(N, 3, M, 32) * 1.2).astype(np.float32)
    fd = (rng.randn(N, 3, M, 32) * 0.9 + 0.2).astype(np.float32)
    data = dict(features_norm=fn, features_denorm=fd)
    dis = orc.random_decoder(seed + 1, bias_scale=0.3)
    out = ref_decoder(dis)(torch.from_numpy(fn), torch.from_numpy(fd), None)
    data.update({"dis." + k: v for k, v in dis.items()}, **{"dis.out." + k: v.numpy() for k, v in out.items()})
    for k, v in zip(("rgb", "sigma", "seg"), orc.decoder_disentangled(fn, fd, dis)):
        check("dis." + k, v, out[k].numpy(), 2e-5)
    seg = orc.random_segmentation_decoder(seed + 2, bias_scale=0.3)
    m = SegmentationOSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32, "decoder_seg_dim": 15})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seg.items()})
    out = m.eval()(torch.from_numpy(fn), torch.from_numpy(fd), None)
    data.update({"seg." + k: v for k, v in seg.items()}, **{"seg.out." + k: v.numpy() for k, v in out.items()})
    for k, v in zip(("rgb", "sigma", "seg"), orc.decoder_segmentation(fd, seg)):
        check("seg." + k, v, out[k].numpy(), 2e-5)
    w = {"net.0.weight": rng.randn(64, 32), "net.0.bias": rng.randn(64) * 0.3, "net.2.weight": rng.randn(33, 64), "net.2.bias": rng.randn(33) * 0.3}
    w = {k: v.astype(np.float32) for k, v in w.items()}
    m = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    out = m.eval()(torch.from_numpy(fd), None)
    data.update({"osg." + k: v for k, v in w.items()}, **{"osg.out." + k: v.numpy() for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "decoder_forward.npz"), **data)
    print("  wrote decoder_forward.npz")


def gen_point_query(seed=7, N=2, H=16, P=500):
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    dec_np = orc.random_decoder(seed + 1, bias_scale=0.3)
    coords = (rng.rand(N, P, 3).astype(np.float32) - 0.5) * 1.3          # some points outside the box
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)
    tp = torch.from_numpy(planes)
    norm, _, _ = TriPlaneGenerator.normalize_plane(G, tp)
    rend = DisentangledImportanceRenderer()
    out = rend.run_model(norm.view(N, 3, 32, H, H), tp.view(N, 3, 32, H, H), ref_decoder(dec_np),
                         torch.from_numpy(coords), None, {"box_warp": 1})
    ref = {k: v.numpy() for k, v in out.items()}
    mine = orc.point_query(planes, dec_np, coords, {"box_warp": 1})
    for k in ref:
        check("pq." + k, mine[k], ref[k], 2e-5)
    np.savez_compressed(os.path.join(OUT, "point_query.npz"), planes=planes, coords=coords,
                        **{"dec." + k: v for k, v in dec_np.items()}, **{"out." + k: v for k, v in ref.items()})
    print("  wrote point_query.npz")


def gen_plane_stats(seed=11, N=3, H=12):
    rng = np.random.RandomState(seed)
    planes = smooth_planes(rng, N, H)
    G = TriPlaneGenerator.__new__(TriPlaneGenerator)
    tp = torch.from_numpy(planes)
    norm, mean, std = TriPlaneGenerator.normalize_plane(G, tp)
    ext_mean = torch.from_numpy(rng.randn(1, 96, 1, 1).astype(np.float32))
    ext_std = torch.from_numpy(np.abs(rng.randn(1, 96, 1, 1)).astype(np.float32) + 0.1)
    den_t = TriPlaneGenerator.denormalize_plane(G, norm, ext_mean, ext_std)
    den_i = TriPlaneGenerator.denormalize_plane(G, norm, mean[1][None], std[2][None])     # triplane.py:100-101
    on, omean, ostd = orc.normalize_plane(planes)
    check("mean", omean, mean.numpy(), 1e-6)
    check("std", ostd, std.numpy(), 1e-6)
    check("norm", on, norm.numpy(), 2e-5)
    _, d_t, _, _ = orc.synthesis_planes(planes, ext_mean.numpy(), ext_std.numpy())
    _, d_i, _, _ = orc.synthesis_planes(planes, 1, 2)
    check("denorm_tensor", d_t.reshape(N, 96, H, H), den_t.numpy(), 2e-5)
    check("denorm_int", d_i.reshape(N, 96, H, H), den_i.numpy(), 2e-5)
    np.savez_compressed(os.path.join(OUT, "plane_stats.npz"), planes=planes, mean=mean.numpy(), std=std.numpy(),
                        norm=norm.numpy(), ext_mean=ext_mean.numpy(), ext_std=ext_std.numpy(),
                        denorm_tensor=den_t.numpy(), denorm_int_1_2=den_i.numpy())
    print("  wrote plane_stats.npz")


def gen_ray_sampler():
    # gen_samples.py:166 yaw/pitch triples + a skewed, off-centre intrinsics case; gen_videos.py:128-133 orbit
    c2w, K = cams([(0.4, -0.2), (0.0, -0.2), (-0.4, -0.2)])
    K2 = K.clone()
    K2[1, 0, 1] = 0.3
    K2[1, 0, 2] = 0.45
    K2[1, 1, 1] = 3.9
    K2[2, 1, 2] = 0.55
    orbit = []
    for f in (0, 1, 7, 100):
        orbit.append(LookAtPoseSampler.sample(3.14 / 2 + 0.35 * np.sin(2 * 3.14 * f / 240),
                                              3.14 / 2 - 0.05 + 0.25 * np.cos(2 * 3.14 * f / 240),
                                              torch.tensor([0, 0, 0.2]), radius=2.7))
    orbit = torch.cat(orbit, 0)
    Ko = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]])[None].repeat(4, 1, 1)
    data = {}
    for tag, (cc, kk, R) in dict(a=(c2w, K, 8), b=(c2w, K2, 12), orbit=(orbit, Ko, 6)).items():
        o, d = RaySampler()(cc, kk, R)
        oo, od = orc.ray_sampler(cc.numpy(), kk.numpy(), R)
        check(f"rs.{tag}.o", oo, o.numpy(), 1e-6)
        check(f"rs.{tag}.d", od, d.numpy(), 1e-6)
        data.update({f"{tag}.cam2world": cc.numpy(), f"{tag}.intrinsics": kk.numpy(), f"{tag}.R": R,
                     f"{tag}.origins": o.numpy(), f"{tag}.dirs": d.numpy()})
    # the oracle's own camera helpers against the reference's
    for (y, p) in [(0.4, -0.2), (-0.4, -0.2), (0.0, 0.0)]:
        a = LookAtPoseSampler.sample(math.pi / 2 + y, math.pi / 2 + p, torch.tensor([0, 0, 0.2]), radius=2.7).numpy()
        b = orc.lookat_pose(math.pi / 2 + y, math.pi / 2 + p, [0, 0, 0.2], radius=2.7)
        check("lookat", b, a, 1e-6)
    check("fov", orc.fov_to_intrinsics(18.837), FOV_to_intrinsics(18.837).numpy(), 1e-7)
    np.savez_compressed(os.path.join(OUT, "ray_sampler.npz"), **data)
    print("  wrote ray_sampler.npz")


def gen_camera_samples():
    """utils.get_camera_samples (utils.py:130-144) and the camera schedule of utils.render_video (:45-73), by importing the
    reference's utils.py itself.  It imports imageio and torchvision (absent here, used only by the mp4 writer / make_grid):
    empty placeholder modules let the import proceed; render_video is run with a recording stand-in for the writer and for
    decode(), so the (pitch, yaw) list it visits is captured from the reference's own loop."""
    import types
    io = types.ModuleType("imageio")
    tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
    tvu.make_grid = None
    tv.utils = tvu
    for name, mod in (("imageio", io), ("torchvision", tv), ("torchvision.utils", tvu)):
        sys.modules.setdefault(name, mod)
    import utils as ref_utils
    G = types.SimpleNamespace(rendering_kwargs={"avg_camera_pivot": [0, 0, 0.2], "avg_camera_radius": 2.7})
    cams_ref = torch.cat(ref_utils.get_camera_samples(G, torch.device("cpu")), 0).numpy()
    G0 = types.SimpleNamespace(rendering_kwargs={})
    cams_default = torch.cat(ref_utils.get_camera_samples(G0, torch.device("cpu")), 0).numpy()
    seen = []

    class Writer:
        def append_data(self, img): pass
        def close(self): pass
    io.get_writer = lambda fn, **k: Writer()
    real_decode = ref_utils.decode
    ref_utils.decode = lambda G_, ws, cam, n, d, **k: (seen.append(cam.numpy().copy()), {"image": torch.zeros(1, 3, 2, 2)})[1]
    try:
        ref_utils.render_video(G, "/tmp/_nfe_golden/x.mp4", torch.zeros(1, 14, 512), None, None, frames=12, a_degree=15.0, b_degree=12.0)
        video = np.concatenate(seen, 0)
        del seen[:]
        ref_utils.render_video(G0, "/tmp/_nfe_golden/x.mp4", torch.zeros(1, 14, 512), None, None, frames=9, a_degree=10.0, b_degree=20.0,
                               init_pitch=1.2, init_yaw=1.7)
        video2 = np.concatenate(seen, 0)
    finally:
        ref_utils.decode = real_decode
    np.savez_compressed(os.path.join(OUT, "camera_samples.npz"), cams_pivot02=cams_ref, cams_default=cams_default,
                        video_cams_12=video, video_cams_9_interp=video2)
    print(f"  wrote camera_samples.npz ({cams_ref.shape}, video {video.shape})")


def main():
    os.makedirs(OUT, exist_ok=True)
    front3 = [(0.4, -0.2), (0.0, -0.2), (-0.4, -0.2)]
    print("render core:")
    gen_render_case("single_r8_d8", 1, 2, 8, 16, 8, 0, front3[:2])
    gen_render_case("single_r16_d48", 2, 1, 16, 32, 48, 0, front3[:1])
    gen_render_case("two_r8_d8_i8", 3, 2, 8, 16, 8, 8, front3[1:])
    gen_render_case("two_r16_d48_i48", 4, 2, 16, 24, 48, 48, front3[:2])
    gen_render_case("two_swap_white", 5, 2, 8, 16, 12, 12, front3[::2], white_back=True, swap=True)
    # rays leaving the box: box_warp<1 makes most samples fall outside the planes (zeros padding)
    gen_render_case("oob_boxwarp", 6, 1, 8, 16, 16, 16, [(0.9, 0.3)], box_warp=0.6)
    gen_render_case("disparity", 8, 1, 8, 16, 10, 6, front3[:1], disparity=True)
    gen_render_case("auto_limits", 9, 2, 8, 16, 12, 12, [(0.4, -0.2), (1.2, 0.5)], auto=True)
    print("render core, real sizes:")
    gen_render_full_size()
    gen_render_full_size("ffhq_render", seed=411, N=2, R=128, H=256, D=48, Ni=48, swap=True, stride=7, chunk=16384,
                         angles=((0.35, -0.15), (-0.3, 0.1)))
    gen_render_full_size("cfg5_render", seed=421, N=2, R=128, H=256, D=96, Ni=96, swap=True, stride=7, chunk=8192,
                         angles=((0.3, -0.1), (-0.25, 0.15)))
    gen_render_full_size("cfg5_render_ws", seed=431, N=2, R=512, H=256, D=96, Ni=96, swap=True, stride=127, chunk=8192,
                         angles=((0.3, -0.1), (-0.25, 0.15)))
    gen_legacy_renderer()
    gen_segmentation_decoder()
    gen_decoder_forward()
    print("point query:")
    gen_point_query()
    print("plane stats:")
    gen_plane_stats()
    print("ray sampler:")
    gen_ray_sampler()
    gen_camera_samples()
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"total fixture size {sz/1e6:.2f} MB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "render_full_size":       # one fixture only
        os.makedirs(OUT, exist_ok=True)
        gen_render_full_size()
    elif len(sys.argv) > 1 and sys.argv[1] == "legacy_renderer":
        os.makedirs(OUT, exist_ok=True)
        gen_legacy_renderer()
    elif len(sys.argv) > 1 and sys.argv[1] == "segmentation_decoder":
        os.makedirs(OUT, exist_ok=True)
        gen_segmentation_decoder()
    elif len(sys.argv) > 1 and sys.argv[1] == "camera_samples":
        os.makedirs(OUT, exist_ok=True)
        gen_camera_samples()
    elif len(sys.argv) > 1 and sys.argv[1] == "decoder_forward":
        os.makedirs(OUT, exist_ok=True)
        gen_decoder_forward()
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg5_render":
        # BASELINE config 5: 96 + 96 samples (projector.py:33-34), two plane sets with swapped statistics (utils.py:176 path)
        os.makedirs(OUT, exist_ok=True)
        gen_render_full_size("cfg5_render", seed=421, N=2, R=128, H=256, D=96, Ni=96, swap=True, stride=7, chunk=8192,
                             angles=((0.3, -0.1), (-0.25, 0.15)))
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg5_render_ws":
        os.makedirs(OUT, exist_ok=True)
        gen_render_full_size("cfg5_render_ws", seed=431, N=2, R=512, H=256, D=96, Ni=96, swap=True, stride=127, chunk=8192,
                             angles=((0.3, -0.1), (-0.25, 0.15)))
    elif len(sys.argv) > 1 and sys.argv[1] == "ffhq_render":
        os.makedirs(OUT, exist_ok=True)
        gen_render_full_size("ffhq_render", seed=411, N=2, R=128, H=256, D=48, Ni=48, swap=True, stride=7, chunk=16384,
                             angles=((0.35, -0.15), (-0.3, 0.1)))
    else:
        main()
