"""GPU: the whole TriPlaneGenerator path (mapping -> backbone -> plane statistics -> fused render -> SR)
through the reference's own call signatures, against golden vectors captured from the reference
TriPlaneGenerator (reduced backbone width; full-size SR head and decoder).  Bar: <= 1e-3 max-abs
(BASELINE.json north_star) with identical latents, poses and jitter."""
import numpy as np
import pytest
import torch

from oracle.dense_params import generator_params
from tests._golden import load

pytestmark = pytest.mark.gpu
TOL = 1e-3

RENDERING_KWARGS = dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True,
                        c_gen_conditioning_zero=False, c_scale=1, superresolution_noise_mode="none", depth_resolution=12,
                        depth_resolution_importance=12, ray_start=2.25, ray_end=3.3, box_warp=1,
                        disparity_space_sampling=False, clamp_mode="softplus", decoder_lr_mul=1)


@pytest.fixture(scope="module")
def setup():
    from nerffaceediting_amd.training.triplane import TriPlaneGenerator
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    z = load("dense_e2e")
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2),
                          rendering_kwargs=dict(RENDERING_KWARGS),
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=4096, channel_max=32, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    sd = G.state_dict()
    for k, v in generator_params(int(z["seed"]), 4096, 32).items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = v
    G.load_state_dict(sd)
    return G.to(dev).eval().requires_grad_(False), z, dev


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def err(a, b):
    return float((a.detach().cpu().double() - torch.from_numpy(np.asarray(b)).double()).abs().max())


def test_mapping(setup):
    G, z, dev = setup
    ws = G.mapping(t(z["z"], dev), t(z["c"], dev), truncation_psi=0.7, truncation_cutoff=14)
    assert err(ws, z["ws"]) <= 1e-4


@pytest.mark.parametrize("tag", ["plain", "swap"])
def test_synthesis(setup, tag):
    G, z, dev = setup
    R = int(z["R"])
    kw = dict(planes_mean=1, planes_var=0) if tag == "swap" else {}        # (int,int) override, triplane.py:100-101
    G.renderer.inject_jitter(t(z["u_coarse"], dev), t(z["u_fine"], dev))
    out = G.synthesis(t(z["ws"], dev), t(z["c"], dev), neural_rendering_resolution=R, noise_mode="const", **kw)
    assert G.neural_rendering_resolution == R                                  # stored on self, triplane.py:78-81
    assert out["image"].shape == (2, 3, 512, 512) and out["image_seg"].shape == (2, 15, R, R)
    assert out["image_raw"].shape == (2, 3, R, R) and out["image_depth"].shape == (2, 1, R, R)
    errs = {"image": err(out["image"][:, :, ::4, ::4], z[tag + ".image_s4"])}
    for k in ("image_seg", "image_raw", "image_depth", "plane_mean", "plane_var"):
        errs[k] = err(out[k], z[f"{tag}.{k}"])
    print(tag, errs)
    for k, e in errs.items():
        assert e <= TOL, (k, e)
    assert abs(float(out["image"].mean()) - float(z[tag + ".image_mean"])) <= 1e-4


def test_forward_and_cached_backbone(setup):
    G, z, dev = setup
    R = int(z["R"])
    uc, uf = t(z["u_coarse"], dev), t(z["u_fine"], dev)
    G.renderer.inject_jitter(uc, uf)
    a = G(t(z["z"], dev), t(z["c"], dev), truncation_psi=0.7, truncation_cutoff=14, neural_rendering_resolution=R,
          noise_mode="const", cache_backbone=True)
    assert err(a["image_raw"], z["plain.image_raw"]) <= TOL
    G.renderer.inject_jitter(uc, uf)
    b = G.synthesis(torch.zeros_like(t(z["ws"], dev)), t(z["c"], dev), use_cached_backbone=True, noise_mode="const")
    # the planes came from the cache, not from the zero ws (the SR head still sees ws, so compare the raw render)
    for k in ("image_raw", "image_seg", "image_depth", "plane_mean"):
        assert torch.equal(a[k], b[k]), k
    G._last_planes = None


def test_sample_mixed(setup):
    G, z, dev = setup
    out = G.sample_mixed(t(z["sample.coords"], dev), None, t(z["ws"], dev), noise_mode="const")
    for k in ("rgb", "sigma", "seg"):
        assert err(out[k], z["sample." + k]) <= TOL, k


def test_render_views_and_density_grid(setup):
    """apps.render_views (batched counterpart of gen_samples/gen_videos loops) and extract_density."""
    from nerffaceediting_amd import apps
    G, z, dev = setup
    ws = t(z["ws"], dev)[:1]
    c = apps.sample_cameras(dev)
    assert c.shape == (3, 25)
    G.neural_rendering_resolution = 32
    torch.manual_seed(1)
    a = apps.render_views(G, ws, c, batch=2, noise_mode="const")
    torch.manual_seed(1)
    b = apps.render_views(G, ws, c, batch=3, noise_mode="const")
    assert a.shape == (3, 3, 512, 512) and torch.isfinite(a).all()
    assert apps.to_uint8(a).dtype == torch.uint8
    oc = apps.orbit_cameras(8, dev)
    assert oc.shape == (8, 25) and float((oc[0] - oc[4]).abs().max()) > 1e-3
    sig = apps.extract_density(G, ws, shape_res=24, max_batch=5000, noise_mode="const")
    # off-diagonal probes: sig[i,j,k] is the density at the create_samples point of flat index (i*R + j)*R + k
    pts, _, _ = apps.create_samples(N=24, cube_length=1.0, device=dev)
    probe = torch.tensor([(1 * 24 + 2) * 24 + 3, (20 * 24 + 5) * 24 + 11, 0, 24 ** 3 - 1], device=dev)
    want = G.sample_mixed(pts[:, probe].contiguous(), None, ws, noise_mode="const")["sigma"].reshape(-1)
    got = torch.stack([sig[1, 2, 3], sig[20, 5, 11], sig[0, 0, 0], sig[-1, -1, -1]])
    assert float((got - want).abs().max()) <= 1e-5
    del b


def test_stream_ring_matches_single_stream(setup):
    """apps.render_views on two alternating HIP streams (StreamRing) == the same batches on one stream, bit for bit (same Philox
    keys: the seeds are drawn from torch's CPU generator in call order), uint8 conversion included."""
    from nerffaceediting_amd import apps
    G, z, dev = setup
    ws = t(z["ws"], dev)[:1]
    c = apps.orbit_cameras(5, dev)
    old = G.neural_rendering_resolution
    G.neural_rendering_resolution = 32
    try:
        outs = []
        for streams in (1, 2, 3):
            torch.manual_seed(9)
            outs.append(apps.render_views(G, ws, c, batch=2, streams=streams, noise_mode="const"))
        torch.manual_seed(9)
        u8 = apps.render_views(G, ws, c, batch=2, streams=2, uint8=True, noise_mode="const")
    finally:
        G.neural_rendering_resolution = old
    assert outs[0].shape == (5, 3, 512, 512)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(u8, apps.to_uint8(outs[0]))


def test_density_grid_vs_reference(setup):
    """gen_samples.py's shape sweep (create_samples grid, chunked G.sample, reshape) on a 16^3 grid against the reference
    generator's own sweep; then the flip + border trim that precedes the .mrc / marching-cubes export."""
    from nerffaceediting_amd import apps
    G, z, dev = setup
    g = load("density_grid")
    R = int(g["shape_res"])
    sig = apps.extract_density(G, t(z["ws"], dev)[:1], shape_res=R, max_batch=int(g["max_batch"]), noise_mode="const")
    assert sig.shape == (R, R, R)
    e = err(sig, g["sigma_grid"])
    print("density grid", e, float(np.abs(g["sigma_grid"]).max()))
    assert e <= TOL
    assert err(apps.density_to_volume(sig), g["sigma_volume"]) <= TOL


def test_graphed_synthesis_matches_eager(setup):
    """hipGraph replay == eager synthesis (same Philox key), and new inputs take effect on replay."""
    from nerffaceediting_amd.graphs import GraphedSynthesis
    G, z, dev = setup
    ws, c = t(z["ws"], dev), t(z["c"], dev)
    g = GraphedSynthesis(G, batch=2, neural_rendering_resolution=32, noise_mode="const")
    seed = 777
    out = {k: v.clone() for k, v in g(ws, c, seed=seed).items()}
    G.renderer.seed_tensor = torch.tensor([seed], dtype=torch.int64, device=dev)
    try:
        eager = G.synthesis(ws, c, neural_rendering_resolution=32, noise_mode="const")
    finally:
        G.renderer.seed_tensor = None
    for k in ("image", "image_raw", "image_seg", "image_depth"):
        assert torch.equal(out[k], eager[k]), k
    out2 = {k: v.clone() for k, v in g(ws.flip(0), c.flip(0), seed=seed).items()}        # new inputs take effect
    G.renderer.seed_tensor = torch.tensor([seed], dtype=torch.int64, device=dev)
    try:
        eager2 = G.synthesis(ws.flip(0).contiguous(), c.flip(0).contiguous(), neural_rendering_resolution=32, noise_mode="const")
    finally:
        G.renderer.seed_tensor = None
    assert torch.equal(out2["image"], eager2["image"]) and not torch.equal(out2["image"], out["image"])
    out3 = g(ws, c, seed=seed + 1)
    assert not torch.equal(out3["image_raw"], out["image_raw"])


def test_demo_helpers_encode_decode(setup):
    """utils.encode -> normalize -> decode reproduces synthesis(); swapping statistics between the two
    identities reproduces the (int,int) appearance override."""
    from nerffaceediting_amd import utils as U
    G, z, dev = setup
    ws, c = t(z["ws"], dev), t(z["c"], dev)
    uc, uf = t(z["u_coarse"], dev), t(z["u_fine"], dev)
    G.neural_rendering_resolution = int(z["R"])
    planes = U.encode(G, ws, noise_mode="const")
    assert planes.shape == (2, 3, 32, 256, 256)
    norm, mean, var = U.normalize_plane(planes)
    assert err(mean.reshape(2, 96, 1, 1), z["plain.plane_mean"]) <= 1e-4 and err(var.reshape(2, 96, 1, 1), z["plain.plane_var"]) <= 1e-4
    G.renderer.inject_jitter(uc, uf)
    out = U.decode(G, ws, c, norm, planes, noise_mode="const")
    for k in ("image_raw", "image_seg", "image_depth"):
        assert err(out[k], z["plain." + k]) <= TOL, k
    assert err(out["image"][:, :, ::4, ::4], z["plain.image_s4"]) <= TOL
    swapped = U.denormalize_plane(norm, mean[1:2], var[0:1])              # == planes_mean=1, planes_var=0
    G.renderer.inject_jitter(uc, uf)
    out = U.decode(G, ws, c, norm, swapped, noise_mode="const")
    assert err(out["image_raw"], z["swap.image_raw"]) <= TOL
    # default start pose == orbit start (pi/2 - 15deg == 5pi/12): no lead-in frames; another init pose adds frames//4
    frames = U.render_video_frames(G, ws[:1], norm[:1], planes[:1], frames=4, batch=3)
    assert frames.shape == (4, 512, 512, 3) and frames.dtype == torch.uint8
    assert len(U.video_camera_schedule(8, init_pitch=1.0)) == 10


def test_render_video_reference_signature(setup, tmp_path):
    """utils.render_video(G, fn, ws, norm_planes, denorm_planes, frames, fps, ...) (utils.py:32-88): positional signature of the
    reference; frames go to a caller-supplied writer (imageio's protocol or a callable) or a .npy file."""
    from nerffaceediting_amd import utils as U
    G, z, dev = setup
    ws = t(z["ws"], dev)[:1]
    G.neural_rendering_resolution = int(z["R"])
    planes = U.encode(G, ws, noise_mode="const")
    norm, _, _ = U.normalize_plane(planes)

    class Writer:
        def __init__(self): self.frames, self.closed = [], False
        def append_data(self, f): self.frames.append(f)
        def close(self): self.closed = True
    torch.manual_seed(5)
    want = U.render_video_frames(G, ws, norm, planes, frames=8, a_degree=10.0, b_degree=8.0, batch=2)
    w = Writer()
    torch.manual_seed(5)
    got = U.render_video(G, str(tmp_path / "sub" / "v.mp4"), ws, norm, planes, 8, 30, 10.0, 8.0, writer=w, batch=2)
    assert len(w.frames) == 10 and not w.closed      # 8 orbit frames + 8 // 4 lead-in frames (utils.py:45-66)
    assert w.frames[0].shape == (512, 512, 3) and w.frames[0].dtype == np.uint8
    assert torch.equal(got, want) and np.array_equal(np.stack(w.frames), want.cpu().numpy())
    seen = []
    torch.manual_seed(5)
    U.render_video(G, None, ws, norm, planes, frames=8, a_degree=10.0, b_degree=8.0, writer=seen.append, batch=2)
    assert np.array_equal(np.stack(seen), want.cpu().numpy())
    fn = str(tmp_path / "v.npy")
    torch.manual_seed(5)
    U.render_video(G, fn, ws, norm, planes, frames=8, a_degree=10.0, b_degree=8.0, batch=2)
    assert np.array_equal(np.load(fn), want.cpu().numpy())
    cams = U.get_camera_samples(G, dev)
    assert len(cams) == 9 and cams[0].shape == (1, 25) and cams[0].device.type == "cuda"
    out = U.decode(G, ws, torch.cat(cams[:2], 0), norm, planes, noise_mode="const")
    assert out["image"].shape == (2, 3, 512, 512)


def test_disable_disentangle_ablation(setup):
    """disable_disentangle=True (triplane.py:93,104-107,119): both heads read the raw planes; equals the renderer called
    with norm_planes = denorm_planes = raw planes, and returns no plane statistics."""
    from nerffaceediting_amd.training.triplane import TriPlaneGenerator
    G, z, dev = setup
    kw = dict(G.init_kwargs); kw["disable_disentangle"] = True
    G2 = TriPlaneGenerator(*G.init_args, **kw).to(dev).eval().requires_grad_(False)
    G2.load_state_dict(G.state_dict())
    ws, c = t(z["ws"], dev), t(z["c"], dev)
    uc, uf = t(z["u_coarse"], dev), t(z["u_fine"], dev)
    R = int(z["R"])
    G2.renderer.inject_jitter(uc, uf)
    out = G2.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const", planes_mean=1, planes_var=0)   # override ignored, :93
    assert out["plane_mean"] is None and out["plane_var"] is None
    planes = G.backbone.synthesis(ws, noise_mode="const").view(2, 3, 32, 256, 256)
    o, d = G.ray_sampler(c[:, :16].reshape(-1, 4, 4), c[:, 16:25].reshape(-1, 3, 3), R)
    G.renderer.inject_jitter(uc, uf)
    feat, seg, depth, _ = G.renderer(planes, planes, G.decoder, o, d, G.rendering_kwargs)
    assert err(out["image_raw"], feat[..., :3].permute(0, 2, 1).reshape(2, 3, R, R).cpu().numpy()) <= 1e-4
    assert err(out["image_seg"], seg.permute(0, 2, 1).reshape(2, 15, R, R).cpu().numpy()) <= 1e-4
    assert err(out["image_raw"], z["plain.image_raw"]) > 1e-2                     # and it differs from the disentangled render
    sig = G2.sample_mixed(t(z["sample.coords"], dev), None, ws, noise_mode="const")["sigma"]
    assert torch.isfinite(sig).all()


def test_full_size_generator_forward():
    """The FFHQ-size generator (30.7 M parameters, full-width backbone and SR head) through forward(): mapping -> backbone
    -> statistics -> 64^2 x (24+24) render -> SR, against outputs captured from the reference TriPlaneGenerator."""
    from nerffaceediting_amd.training.triplane import TriPlaneGenerator
    dev = torch.device("cuda:0")
    z = load("dense_e2e_full")
    rk = dict(RENDERING_KWARGS, depth_resolution=int(z["D"]), depth_resolution_importance=int(z["Di"]))
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=32768, channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    sd = G.state_dict()
    for k, v in generator_params(int(z["seed"]), 32768, 512).items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = v
    G.load_state_dict(sd)
    G = G.to(dev).eval().requires_grad_(False)
    assert sum(p.numel() for p in G.parameters()) == 30665223
    G.renderer.inject_jitter(t(z["u_coarse"], dev), t(z["u_fine"], dev))
    out = G(t(z["z"], dev), t(z["c"], dev), truncation_psi=0.7, truncation_cutoff=14, neural_rendering_resolution=int(z["R"]),
            noise_mode="const")
    errs = {"image": err(out["image"][:, :, 1::4, 2::4], z["image_s4"])}
    for k in ("image_seg", "image_raw", "image_depth", "plane_mean", "plane_var"):
        errs[k] = err(out[k], z[k])
    print("full-size forward", errs)
    for k, e in errs.items():
        assert e <= TOL, (k, e)
    assert abs(float(out["image"].mean()) - float(z["image_mean"])) <= 1e-4


def _full_generator(dev, seed, D, Di):
    from nerffaceediting_amd.training.triplane import TriPlaneGenerator
    rk = dict(RENDERING_KWARGS, depth_resolution=D, depth_resolution_importance=Di)
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=32768, channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    sd = G.state_dict()
    for k, v in generator_params(seed, 32768, 512).items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = v
    G.load_state_dict(sd)
    return G.to(dev).eval().requires_grad_(False)


def test_full_size_generator_cfg1_exact():
    """BASELINE config 1 exactly: the FFHQ-size generator through forward(), one latent
    (RandomState(0)), frontal camera, 64^2 neural render x (48 + 48) samples, against the reference TriPlaneGenerator."""
    dev = torch.device("cuda:0")
    z = load("dense_e2e_cfg1")
    R, D, Di = int(z["R"]), int(z["D"]), int(z["Di"])
    assert (R, D, Di) == (64, 48, 48)
    G = _full_generator(dev, int(z["seed"]), D, Di)
    rng = np.random.RandomState(int(z["u_seed"]))                     # oracle/gen_golden_dense.py gen_e2e_cfg1
    u_c = rng.rand(1, R * R, D).astype(np.float32)
    u_f = rng.rand(R * R, Di).astype(np.float32)
    assert np.array_equal(z["z"], np.random.RandomState(0).randn(1, 512).astype(np.float32))
    G.renderer.inject_jitter(t(u_c, dev), t(u_f, dev))
    out = G(t(z["z"], dev), t(z["c"], dev), neural_rendering_resolution=R, noise_mode="const")
    errs = {"image": err(out["image"][:, :, 1::4, 2::4], z["image_s4"])}
    for k in ("image_seg", "image_raw", "image_depth", "plane_mean", "plane_var"):
        errs[k] = err(out[k], z[k])
    print("cfg1 forward", errs)
    for k, e in errs.items():
        assert e <= TOL, (k, e)
    assert abs(float(out["image"].double().mean()) - float(z["image_mean"])) <= 1e-4


# fp16 conv operands (round 4: the reference's own GPU arithmetic for its fp16 layers): 2 x the errors measured on MI355X against the
# fp32 CPU capture (profiles/r04_fp16_error.md)
FP16_BOUND = {"image": (0.0027, 0.0006), "image_raw": (0.00027, 4.5e-5), "image_seg": (0.0004, 6.5e-5), "image_depth": (1e-4, 1.2e-5)}


@pytest.mark.parametrize("conv_math", ["bf16x3", "bf16", "fp16"])
def test_full_size_synthesis_cfg3(conv_math):
    """BASELINE config 3's data path at full size: synthesis() at neural_rendering_resolution 512 x 64 samples, so the 32-channel
    512^2 feature image goes through the antialiased 512 -> 128 down-resize into the SR head (superresolution.py:279-290),
    against the reference TriPlaneGenerator (one view).  bf16x3 (fp32-grade convs): the 1e-3 bar.  bf16 (the throughput mode
    bench.py --workload full times): bounds relative to each output's magnitude, since plain-bf16 operands carry 2^-9
    relative rounding through 13 + 6 conv layers."""
    dev = torch.device("cuda:0")
    z = load("dense_e2e_cfg3")
    R, D = int(z["R"]), int(z["D"])
    assert (R, D, int(z["Di"])) == (512, 64, 0)
    G = _full_generator(dev, int(z["seed"]), D, 0)
    G.backbone.synthesis.conv_math = conv_math
    G.superresolution.conv_math = conv_math
    u_c = np.random.RandomState(int(z["u_seed"])).rand(1, R * R, D).astype(np.float32)
    ws = G.mapping(t(z["z"], dev), t(z["c"], dev), truncation_psi=0.7, truncation_cutoff=14)
    assert err(ws, z["ws"]) <= 1e-4
    G.renderer.inject_jitter(t(u_c, dev))
    out = G.synthesis(t(z["ws"], dev), t(z["c"], dev), neural_rendering_resolution=R, noise_mode="const")
    assert out["image"].shape == (1, 3, 512, 512) and out["image_raw"].shape == (1, 3, R, R) and out["image_seg"].shape == (1, 15, R, R)
    errs = {"image": err(out["image"][:, :, 1::4, 2::4], z["image_s4"]),
            "image_raw": err(out["image_raw"][:, :, 1::3, 2::3], z["image_raw_s3"]),
            "image_seg": err(out["image_seg"][:, :, 2::4, 1::4], z["image_seg_s4"]),
            "image_depth": err(out["image_depth"][:, :, ::2, 1::2], z["image_depth_s2"])}
    means = {k: float(np.abs(out[k].double().mean(dim=(0, 2, 3)).cpu().numpy() - z[k + "_mean"]).max())
             for k in ("image", "image_raw", "image_seg", "image_depth")}
    print("cfg3 synthesis", conv_math, errs, "channel means", means)
    if conv_math == "bf16x3":
        for k, e in errs.items():
            assert e <= TOL, (k, e)
        for k, e in means.items():
            assert e <= 1e-4, (k, e)
        assert err(out["plane_mean"], z["plane_mean"]) <= TOL and err(out["plane_var"], z["plane_var"]) <= TOL
    else:       # 2 x the errors measured on MI355X (profiles/r03_bf16_error.md, r04_fp16_error.md): max-abs, then per-channel means
        bound = {"image": (0.025, 0.006), "image_raw": (0.0025, 0.0006), "image_seg": (0.0035, 0.0004), "image_depth": (0.0015, 0.0001)}
        if conv_math == "fp16":
            bound = FP16_BOUND
        for k, e in errs.items():
            assert e <= bound[k][0], (k, e)
        for k, e in means.items():
            assert e <= bound[k][1], (k, e)


def test_converted_checkpoint_renders_reference_outputs(tmp_path):
    """f1 end to end: reference pickle -> tools/convert_checkpoint.convert() (run in the build container; its .json and the
    SHA-256 of every tensor it wrote are the fixture checkpoint_e2e.json) -> checkpoint.load_generator() -> synthesis() ->
    outputs captured from the reference generator (dense_e2e.npz).  The tensors are regenerated from the seed and proven
    identical to the converter's output by the hashes before they are written in the converter's format."""
    import hashlib
    import json
    import os
    from nerffaceediting_amd.checkpoint import load_generator
    from nerffaceediting_amd.training.triplane import TriPlaneGenerator
    dev = torch.device("cuda:0")
    with open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_e2e.json")) as f:
        rec = json.load(f)
    meta = rec["converter_json"]
    state = {k: v.numpy() for k, v in generator_params(int(rec["seed"]), int(rec["channel_base"]), int(rec["channel_max"])).items()}
    skeleton = TriPlaneGenerator(*meta["init_args"], **meta["init_kwargs"]).state_dict()      # resample_filter buffers: constants
    for k, v in skeleton.items():
        state.setdefault(k, v.numpy())
    assert sorted(state) == sorted(rec["sha256"])

    def digest(a):
        a = np.ascontiguousarray(a)
        return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()
    for k, v in state.items():
        assert digest(v) == rec["sha256"][k], k
    prefix = str(tmp_path / "converted")
    np.savez(prefix + ".npz", **state)
    with open(prefix + ".json", "w") as f:
        json.dump(meta, f)
    G = load_generator(prefix, device=dev)
    assert G.neural_rendering_resolution == 32 and G.rendering_kwargs["depth_resolution_importance"] == 12
    z = load("dense_e2e")
    G.renderer.inject_jitter(t(z["u_coarse"], dev), t(z["u_fine"], dev))
    out = G(t(z["z"], dev), t(z["c"], dev), truncation_psi=0.7, truncation_cutoff=14, noise_mode="const")     # R = 32 from the checkpoint
    errs = {"image": err(out["image"][:, :, ::4, ::4], z["plain.image_s4"])}
    for k in ("image_seg", "image_raw", "image_depth", "plane_mean", "plane_var"):
        errs[k] = err(out[k], z["plain." + k])
    print("converted checkpoint", errs)
    for k, e in errs.items():
        assert e <= TOL, (k, e)


def test_last_planes_contract(setup):
    """`_last_planes` is the reference's NCHW tensor (triplane.py:88-89,110): cached AFTER an appearance override, so a later
    use_cached_backbone call keeps the overridden appearance and reports its statistics; a caller may assign its own tensor."""
    G, z, dev = setup
    R = int(z["R"])
    ws, c = t(z["ws"], dev), t(z["c"], dev)
    uc, uf = t(z["u_coarse"], dev), t(z["u_fine"], dev)
    try:
        G.renderer.inject_jitter(uc, uf)
        a = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const", cache_backbone=True, planes_mean=1, planes_var=0)
        lp = G._last_planes
        assert isinstance(lp, torch.Tensor) and lp.shape == (2, 96, 256, 256)
        assert err(a["image_raw"], z["swap.image_raw"]) <= TOL
        # the cached planes are the DENORMALISED ones: their statistics are the override's (mean of identity 1, std of identity 0)
        m, s = G.compute_mean_var(lp)
        assert err(m, np.repeat(z["plain.plane_mean"][1:2], 2, 0)) <= 1e-3 and err(s, np.repeat(z["plain.plane_var"][0:1], 2, 0)) <= 1e-3
        G.renderer.inject_jitter(uc, uf)
        b = G.synthesis(torch.zeros_like(ws), c, use_cached_backbone=True, noise_mode="const")      # no override now: appearance is kept
        assert err(b["image_raw"], z["swap.image_raw"]) <= TOL
        assert err(b["plane_mean"], m.cpu().numpy()) <= 1e-5
        # a caller-assigned tensor (5-D view, as utils.encode returns) is honoured
        planes = G.backbone.synthesis(ws, noise_mode="const")
        G._last_planes = planes.view(2, 3, 32, 256, 256)
        G.renderer.inject_jitter(uc, uf)
        d = G.synthesis(torch.zeros_like(ws), c, use_cached_backbone=True, noise_mode="const")
        assert err(d["image_raw"], z["plain.image_raw"]) <= TOL and err(d["plane_mean"], z["plain.plane_mean"]) <= 1e-4
    finally:
        G._last_planes = None


def test_sample_passes_density_noise(setup):
    """sample()/sample_mixed() hand self.rendering_kwargs to run_model (triplane.py:148,157), so `density_noise` perturbs
    sigma there too (renderer.py:285-286): sigma moves by N(0,1) * density_noise, rgb and seg do not."""
    G, z, dev = setup
    ws, coords = t(z["ws"], dev), t(z["sample.coords"], dev)
    old = G.rendering_kwargs
    try:
        G.rendering_kwargs = dict(old, density_noise=0.5)
        torch.manual_seed(11)
        a = G.sample_mixed(coords, None, ws, noise_mode="const")
        torch.manual_seed(11)
        b = G.sample_mixed(coords, None, ws, noise_mode="const")
        torch.manual_seed(12)
        c2 = G.sample_mixed(coords, None, ws, noise_mode="const")
    finally:
        G.rendering_kwargs = old
    assert torch.equal(a["sigma"], b["sigma"]) and not torch.equal(a["sigma"], c2["sigma"])
    for k in ("rgb", "seg"):
        assert err(a[k], z["sample." + k]) <= TOL
    dz = (a["sigma"].cpu().double() - torch.from_numpy(z["sample.sigma"]).double()).flatten() / 0.5
    assert abs(float(dz.mean())) < 0.2 and 0.8 < float(dz.std()) < 1.2, (float(dz.mean()), float(dz.std()))


def test_interpolation_video_frames(setup):
    """gen_videos.gen_interp_video counterpart: keyframes from seeds, cubic interpolation in w, orbit cameras."""
    from nerffaceediting_amd import apps
    G, z, dev = setup
    old = G.neural_rendering_resolution
    G.neural_rendering_resolution = 32
    try:
        G.renderer.seed_tensor = torch.tensor([4242], dtype=torch.int64, device=dev)      # one Philox key for every call below
        frames = apps.interpolation_video_frames(G, [0, 1], w_frames=2, batch=3)
        assert frames.shape == (4, 512, 512, 3) and frames.dtype == torch.uint8
        c = apps.orbit_cameras(4, dev)
        zs = torch.cat([apps.seed_to_z(s, 512, dev) for s in (0, 1)], 0)
        c2w = apps.camera_utils.LookAtPoseSampler.sample(3.14 / 2, 3.14 / 2, torch.tensor([0, 0, 0.2], device=dev), radius=2.7, device=dev)
        c_front = torch.cat([c2w.reshape(-1, 16), torch.tensor(apps.FFHQ_INTRINSICS, device=dev).reshape(-1, 9)], 1).repeat(2, 1)
        ws_key = G.mapping(zs, c_front, truncation_psi=1.0, truncation_cutoff=14)       # gen_videos.py:95-98
        want = apps.to_uint8(G.synthesis(ws_key[:1], c[:1], noise_mode="const")["image"])
        # frame 0 is keyframe 0 under orbit camera 0.  Same Philox key and view index 0 in both calls -> same jitter; the only
        # difference left is the dense kernels' batch-dependent variants (3 views vs 1): at most one uint8 level on a few values
        d = (frames[0].int() - want[0].int()).abs()
        assert int(d.max()) <= 1 and float(d.float().mean()) < 0.01, (int(d.max()), float(d.float().mean()))
        # ... and the interpolated latents are gen_videos.py's: frame 1 = cubic interpolation at t = 1/2 between the keyframes
        ws_all = apps.interpolate_ws(ws_key, w_frames=2)
        want1 = apps.to_uint8(G.synthesis(ws_all[:3].contiguous(), c[:3].contiguous(), noise_mode="const")["image"])
        assert torch.equal(frames[:3], want1)
    finally:
        G.neural_rendering_resolution = old
        G.renderer.seed_tensor = None


def test_decode_is_differentiable_wrt_planes(setup):
    """Plane editing (utils.py:146-199): planes = encode(G, ws); norm/denorm as leaves; decode(); a loss on image_seg /
    image_raw / image_depth back-propagates to the planes.  Checked by a directional finite difference of decode() itself
    (single pass, fixed jitter, fp32 decoder)."""
    from nerffaceediting_amd import utils as U
    G, z, dev = setup
    ws, c = t(z["ws"], dev)[:1], t(z["c"], dev)[:1]
    R = 32
    old_res, old_kw, old_math = G.neural_rendering_resolution, G.rendering_kwargs, G.renderer.decoder_math
    G.neural_rendering_resolution = R
    G.rendering_kwargs = dict(old_kw, depth_resolution=24, depth_resolution_importance=0)
    G.renderer.decoder_math = "fp32"
    try:
        planes = U.encode(G, ws, noise_mode="const")
        norm, mean, var = U.normalize_plane(planes)
        denorm = U.denormalize_plane(norm, mean, var)
        u = torch.rand(1, R * R, 24, device=dev)
        tgt = {k: torch.randn(s, device=dev) for k, s in (("image_seg", (1, 15, R, R)), ("image_raw", (1, 3, R, R)), ("image_depth", (1, 1, R, R)))}

        def loss_of(n_, d_):
            G.renderer.inject_jitter(u)
            out = U.decode(G, ws, c, n_, d_, noise_mode="const")
            return sum((out[k] * tgt[k]).sum() for k in tgt), out
        n1, d1 = norm.clone().requires_grad_(True), denorm.clone().requires_grad_(True)
        loss, out = loss_of(n1, d1)
        assert out["image"].shape == (1, 3, 512, 512)
        out["image"].sum().backward(retain_graph=True)      # R = 32: through the head's pre-resize (its adjoint) as well, since round 3
        assert n1.grad is not None and torch.isfinite(n1.grad).all() and float(n1.grad.abs().max()) > 0
        n1.grad = d1.grad = None
        loss.backward()
        assert n1.grad is not None and d1.grad is not None and torch.isfinite(n1.grad).all() and float(n1.grad.abs().max()) > 0
        # the usual editing loop: only norm_planes is a leaf, the appearance planes are re-derived from it every step
        n2 = norm.clone().requires_grad_(True)
        d2 = U.denormalize_plane(n2, mean, var)
        assert d2.requires_grad
        loss_of(n2, d2)[0].backward()
        both = (n1.grad + d1.grad * var).double()                      # chain rule through denorm = norm * var + mean
        assert float((n2.grad.double() - both).abs().max()) <= 2e-3 * float(both.abs().max())
        with torch.no_grad():
            for which, leaf in ((0, n1), (1, d1)):
                V = torch.randn_like(norm)
                eps = 2e-2
                a = loss_of(norm + eps * V, denorm)[0] if which == 0 else loss_of(norm, denorm + eps * V)[0]
                b = loss_of(norm - eps * V, denorm)[0] if which == 0 else loss_of(norm, denorm - eps * V)[0]
                fd = float(a - b) / (2 * eps)
                an = float((leaf.grad.double() * V.double()).sum())
                assert abs(fd - an) <= 3e-2 * max(abs(an), 1.0), (which, fd, an)
    finally:
        G.neural_rendering_resolution, G.rendering_kwargs, G.renderer.decoder_math = old_res, old_kw, old_math


def test_decode_image_is_differentiable_at_the_head_resolution(setup):
    """utils.decode at neural_rendering_resolution 128 (the FFHQ configuration): out['image'] carries plane gradients through
    the SR head (sr_grad.py) as the reference's decode does by autograd (utils.py:165-199), so an editing loss may mix an image
    term with a segmentation term.  Directional finite difference of decode() itself (fixed jitter, fp32 decoder)."""
    from nerffaceediting_amd import utils as U
    G, z, dev = setup
    ws, c = t(z["ws"], dev)[:1], t(z["c"], dev)[:1]
    R, D = 128, 12
    old_res, old_kw, old_math = G.neural_rendering_resolution, G.rendering_kwargs, G.renderer.decoder_math
    G.neural_rendering_resolution = R
    G.rendering_kwargs = dict(old_kw, depth_resolution=D, depth_resolution_importance=0)
    G.renderer.decoder_math = "fp32"
    try:
        planes = U.encode(G, ws, noise_mode="const")
        norm, mean, var = U.normalize_plane(planes)
        u = torch.rand(1, R * R, D, device=dev)
        gt = torch.Generator(device="cpu").manual_seed(2)
        t_img, t_seg = torch.randn(1, 3, 512, 512, generator=gt).to(dev), torch.randn(1, 15, R, R, generator=gt).to(dev)

        def loss_of(n_):
            G.renderer.inject_jitter(u)
            out = U.decode(G, ws, c, n_, U.denormalize_plane(n_, mean, var), noise_mode="const")
            img_term = (out["image"].double() * t_img.double()).sum()
            return img_term + (out["image_seg"].double() * t_seg.double()).sum(), img_term
        n1 = norm.clone().requires_grad_(True)
        loss, img_term = loss_of(n1)
        g_img, = torch.autograd.grad(img_term, n1, retain_graph=True)          # the image term alone reaches the planes
        assert torch.isfinite(g_img).all() and float(g_img.abs().max()) > 0
        loss.backward()
        with torch.no_grad():
            # direction = the gradient itself at unit RMS (a random direction against random targets gives a derivative near zero,
            # which a central difference across ~1e8 leaky-ReLU kinks and the +-256 clamps of the head cannot resolve); the
            # entry-by-entry check against the reference's autograd is tests/test_sr_grad_gpu.py
            eps = 1e-2
            V = g_img / g_img.pow(2).mean().sqrt()
            fd_img = float(loss_of(norm + eps * V)[1] - loss_of(norm - eps * V)[1]) / (2 * eps)
            an_img = float((g_img.double() * V.double()).sum())
            assert abs(fd_img - an_img) <= 5e-2 * abs(an_img), (fd_img, an_img)
            V = n1.grad / n1.grad.pow(2).mean().sqrt()
            fd = float(loss_of(norm + eps * V)[0] - loss_of(norm - eps * V)[0]) / (2 * eps)
            an = float((n1.grad.double() * V.double()).sum())
            assert abs(fd - an) <= 5e-2 * abs(an), (fd, an)
    finally:
        G.neural_rendering_resolution, G.rendering_kwargs, G.renderer.decoder_math = old_res, old_kw, old_math


def test_plane_optimisation_loop(setup):
    """Geometry editing as an optimisation: make identity A render identity B's parsing map by optimising A's normalised
    planes (appearance statistics untouched); then an appearance edit through the statistics only."""
    from nerffaceediting_amd import editing, utils as U
    G, z, dev = setup
    ws, c = t(z["ws"], dev), t(z["c"], dev)
    old_res, old_kw = G.neural_rendering_resolution, G.rendering_kwargs
    G.neural_rendering_resolution = 32
    G.rendering_kwargs = dict(old_kw, depth_resolution=16, depth_resolution_importance=16)
    try:
        with torch.no_grad():
            planes = U.encode(G, ws, noise_mode="const")
            norm, mean, var = U.normalize_plane(planes)
            target = U.decode(G, ws[1:2], c[:1], norm[1:2], planes[1:2], noise_mode="const")
            labels = target["image_seg"].argmax(1)
        torch.manual_seed(0)
        n2, m2, v2, losses = editing.optimize_planes(G, ws[:1], c[:1], norm[:1], mean[:1], var[:1],
                                                     lambda out: editing.segmentation_loss(out["image_seg"], labels), steps=40, lr=0.05,
                                                     noise_mode="const")
        assert losses[-1] < 0.6 * losses[0], losses[::8]
        assert torch.equal(m2, mean[:1]) and torch.equal(v2, var[:1]) and not torch.equal(n2, norm[:1])
        raw_t = target["image_raw"]
        _, m3, v3, l3 = editing.optimize_planes(G, ws[:1], c[:1], norm[:1], mean[:1], var[:1],
                                                lambda out: (out["image_raw"] - raw_t).square().mean(), steps=25, lr=0.05,
                                                optimize="stats", noise_mode="const")
        assert l3[-1] < l3[0] and not torch.equal(m3, mean[:1])
    finally:
        G.neural_rendering_resolution, G.rendering_kwargs = old_res, old_kw
