"""Golden vectors for the dense half of the path, produced by running the REFERENCE modules on CPU
(build container only: needs /root/reference).  Parameters are drawn from seeded numpy generators
(oracle.dense_params) so fixtures only store seeds, small inputs and outputs.  The torch-CPU oracle
(oracle/dense_oracle.py) is checked against every output before anything is written.

    python oracle/gen_golden_dense.py
"""
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from training import networks_stylegan2 as ref_sg2  # noqa: E402
from training.superresolution import SuperresolutionHybrid8XDC  # noqa: E402

from oracle import dense_oracle as dor  # noqa: E402
from oracle.dense_params import layer_params, mapping_params, sr_params, synthesis_params  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


def check(name, a, b, tol):
    err = float((a.double() - b.double()).abs().max())
    print(f"    oracle vs reference  {name:22s} max-abs {err:.3e}  (|ref| max {float(b.abs().max()):.3g})")
    assert err <= tol, (name, err)


def load(module, params):
    sd = module.state_dict()
    for k, v in params.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k] = v.clone()
    module.load_state_dict(sd)
    return module.eval()


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def gen_mapping():
    z_dim, c_dim, w_dim, num_ws = 64, 25, 64, 6
    p = mapping_params(11, z_dim, c_dim, w_dim, num_layers=2)
    m = load(ref_sg2.MappingNetwork(z_dim, c_dim, w_dim, num_ws, num_layers=2), p)
    rng = np.random.RandomState(12)
    z, c = t(rng.randn(3, z_dim)), t(rng.randn(3, c_dim))
    out = {}
    for tag, (psi, cut) in dict(a=(1.0, None), b=(0.7, None), c=(0.5, 4)).items():
        ref = m(z, c, truncation_psi=psi, truncation_cutoff=cut)
        mine = dor.mapping(p, z, c, num_ws, 2, psi, cut)
        check("mapping." + tag, mine, ref, 1e-5)
        out["ws." + tag] = ref.numpy()
    np.savez_compressed(os.path.join(OUT, "dense_mapping.npz"), z=z.numpy(), c=c.numpy(), seed=11, z_dim=z_dim, c_dim=c_dim,
                        w_dim=w_dim, num_ws=num_ws, **out)
    print("  wrote dense_mapping.npz")


def gen_layers():
    rng = np.random.RandomState(21)
    w_dim = 64
    data = {}
    # SynthesisLayer up=1 with clamp, up=2, ToRGB
    cases = dict(conv_up1=dict(cin=32, cout=48, res=16, up=1, clamp=1.5, gain=1.0),
                 conv_up1_g=dict(cin=16, cout=32, res=12, up=1, clamp=None, gain=0.5),
                 conv_up2=dict(cin=32, cout=32, res=16, up=2, clamp=None, gain=1.0),
                 conv_up2_odd=dict(cin=16, cout=64, res=10, up=2, clamp=4.0, gain=1.0))
    for i, (tag, cfg) in enumerate(cases.items()):
        p = layer_params(100 + i, cfg["cin"], cfg["cout"], w_dim, cfg["res"], k=3)
        m = load(ref_sg2.SynthesisLayer(cfg["cin"], cfg["cout"], w_dim, cfg["res"], up=cfg["up"], conv_clamp=cfg["clamp"]), p)
        rin = cfg["res"] // cfg["up"]
        x, w = t(rng.randn(2, cfg["cin"], rin, rin)), t(rng.randn(2, w_dim))
        ref = m(x, w, noise_mode="const", gain=cfg["gain"])
        mine = dor.synthesis_layer(p, x, w, up=cfg["up"], conv_clamp=cfg["clamp"], gain=cfg["gain"])
        check(tag, mine, ref, 2e-5)
        data.update({f"{tag}.x": x.numpy(), f"{tag}.w": w.numpy(), f"{tag}.out": ref.numpy(),
                     f"{tag}.cfg": np.array([100 + i, cfg["cin"], cfg["cout"], cfg["res"], cfg["up"],
                                             -1 if cfg["clamp"] is None else cfg["clamp"], cfg["gain"]], dtype=np.float64)})
    for i, (tag, cfg) in enumerate(dict(torgb96=dict(cin=32, cout=96, res=16, clamp=None),
                                        torgb3=dict(cin=48, cout=3, res=12, clamp=0.8)).items()):
        p = layer_params(200 + i, cfg["cin"], cfg["cout"], w_dim, cfg["res"], k=1, torgb=True)
        m = load(ref_sg2.ToRGBLayer(cfg["cin"], cfg["cout"], w_dim, conv_clamp=cfg["clamp"]), p)
        x, w = t(rng.randn(2, cfg["cin"], cfg["res"], cfg["res"])), t(rng.randn(2, w_dim))
        ref = m(x, w)
        mine = dor.torgb_layer(p, x, w, conv_clamp=cfg["clamp"])
        check(tag, mine, ref, 2e-5)
        data.update({f"{tag}.x": x.numpy(), f"{tag}.w": w.numpy(), f"{tag}.out": ref.numpy(),
                     f"{tag}.cfg": np.array([200 + i, cfg["cin"], cfg["cout"], cfg["res"], 1,
                                             -1 if cfg["clamp"] is None else cfg["clamp"], 1.0], dtype=np.float64)})
    # upsample2d
    from torch_utils.ops import upfirdn2d
    img = t(rng.randn(2, 5, 6, 6))
    ref = upfirdn2d.upsample2d(img, upfirdn2d.setup_filter([1, 3, 3, 1]))
    check("upsample2d", dor.upsample2d(img), ref, 1e-6)
    data.update({"upsample2d.x": img.numpy(), "upsample2d.out": ref.numpy()})
    # antialiased / plain bilinear resize (superresolution.py:283-286)
    for tag, (hin, hout, aa) in dict(down_aa=(40, 16, True), up_aa=(12, 32, True), down_noaa=(40, 16, False), odd_aa=(50, 16, True)).items():
        x = t(rng.randn(1, 4, hin, hin))
        ref = torch.nn.functional.interpolate(x, size=(hout, hout), mode="bilinear", align_corners=False, antialias=aa)
        check("resize." + tag, dor.resize_bilinear(x, hout, hout, aa), ref, 2e-6)
        data.update({f"resize.{tag}.x": x.numpy(), f"resize.{tag}.out": ref.numpy(), f"resize.{tag}.aa": int(aa)})
    np.savez_compressed(os.path.join(OUT, "dense_layers.npz"), w_dim=w_dim, **data)
    print("  wrote dense_layers.npz")


def gen_synthesis():
    """Reduced backbone: img_resolution 32, channels min(256//res, 32), 96 output channels, w_dim 64."""
    w_dim, res, cb, cm = 64, 32, 256, 32
    net = ref_sg2.SynthesisNetwork(w_dim, res, 96, channel_base=cb, channel_max=cm, num_fp16_res=0, conv_clamp=None,
                                   fused_modconv_default="inference_only")
    p = synthesis_params(31, w_dim, res, 96, cb, cm)
    load(net, p)
    rng = np.random.RandomState(32)
    ws = t(rng.randn(2, net.num_ws, w_dim))
    ref = net(ws, noise_mode="const")
    mine = dor.synthesis_network(p, ws, net.block_resolutions)
    check("synthesis(32px)", mine, ref, 5e-5)
    np.savez_compressed(os.path.join(OUT, "dense_synthesis.npz"), ws=ws.numpy(), out=ref.numpy(), seed=31, w_dim=w_dim, res=res,
                        channel_base=cb, channel_max=cm, num_ws=net.num_ws)
    print("  wrote dense_synthesis.npz")


def gen_synthesis_full():
    """Full-width backbone (train.py FFHQ: 256 px, channels min(32768//res, 512), 96 output channels, w_dim 512), one
    sample.  Only a strided sub-sample of the 25 MB output is stored; it pins the wide-layer kernels (512-channel
    split-K layers, LDS-DMA conv path, chained up-sampling layers) against the reference itself."""
    w_dim, res, cb, cm = 512, 256, 32768, 512
    net = ref_sg2.SynthesisNetwork(w_dim, res, 96, channel_base=cb, channel_max=cm, num_fp16_res=0, conv_clamp=None,
                                   fused_modconv_default="inference_only")
    p = synthesis_params(61, w_dim, res, 96, cb, cm)
    load(net, p)
    rng = np.random.RandomState(62)
    ws = t(rng.randn(1, net.num_ws, w_dim))
    ref = net(ws, noise_mode="const")
    mine = dor.synthesis_network(p, ws, net.block_resolutions)
    check("synthesis(256px, full)", mine, ref, 2e-4 * float(ref.abs().max()))
    np.savez_compressed(os.path.join(OUT, "dense_synthesis_full.npz"), ws=ws.numpy(), out_s8=ref[:, :, 3::8, 5::8].numpy(),
                        ch_mean=ref.mean(dim=(2, 3)).numpy(), ch_std=ref.std(dim=(2, 3)).numpy(), absmax=float(ref.abs().max()),
                        seed=61, w_dim=w_dim, res=res, channel_base=cb, channel_max=cm, num_ws=net.num_ws)
    print("  wrote dense_synthesis_full.npz")


def gen_sr():
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True,
                                   channel_base=32768, channel_max=512, fused_modconv_default="inference_only")
    p = sr_params(41)
    load(sr, p)
    rng = np.random.RandomState(42)
    data = {}
    for tag, r in dict(r64=64, r128=128).items():
        x = t(rng.randn(1, 32, r, r) * 0.5)
        ws = t(rng.randn(1, 14, 512))
        ref = sr(x[:, :3].contiguous(), x, ws, noise_mode="none")
        mine = dor.superresolution_8xdc(p, x[:, :3].contiguous(), x, ws)
        check("sr." + tag, mine, ref, 2e-4)
        data.update({f"{tag}.x": x.numpy(), f"{tag}.ws": ws.numpy(), f"{tag}.out_s4": ref[:, :, ::4, ::4].numpy(),
                     f"{tag}.out_mean": float(ref.mean()), f"{tag}.out_abs_mean": float(ref.abs().mean())})
    np.savez_compressed(os.path.join(OUT, "dense_sr.npz"), seed=41, **data)
    print("  wrote dense_sr.npz")


def gen_sr_backward():
    """Input gradient of the SR head by the reference's own autograd (what utils.decode's differentiable `image` gives,
    utils.py:165-199): SuperresolutionHybrid8XDC at full width, one 128^2 x 32 feature image (rgb = its first 3 channels), a
    seeded cotangent on the 512^2 image -> d<cot, image>/d feature image.  A second case pushes activations into the +-256
    clamp (scaled input) so the clamp masks are exercised.  Stored: the gradient at every second pixel + fp64 channel sums."""
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True,
                                   channel_base=32768, channel_max=512, fused_modconv_default="inference_only")
    load(sr, sr_params(41))
    rng = np.random.RandomState(77)
    data = {}
    with torch.enable_grad():
        # clamped8 (round 4): at scale 40 the +-256 clamps are active on ~2 % of the hidden units but on none of the IMAGE values
        # (VERDICT r3 #8); at scale 100 8.5 % of the image values sit at or beyond 256 (both ToRGB clamps active).  Appended, so the
        # first two cases keep their place in the RandomState(77) sequence.
        for tag, scale in dict(plain=0.5, clamped=40.0, clamped8=100.0).items():
            x = t(rng.randn(1, 32, 128, 128) * scale).requires_grad_(True)
            ws = t(rng.randn(1, 14, 512))
            cot = t(rng.randn(1, 3, 512, 512))
            img = sr(x[:, :3], x, ws, noise_mode="none")
            (g,) = torch.autograd.grad((img * cot).sum(), x)
            frac = float((img.detach().abs() >= 255.999).float().mean())
            print(f"    sr_backward.{tag}: |image| max {float(img.abs().max()):.3g}, clamped share of the image {frac:.4f}, |grad| max {float(g.abs().max()):.3g}")
            # inputs are regenerated by the test from the same RandomState(77) sequence (x, ws, cot per case, in this order)
            data.update({f"{tag}.scale": scale, f"{tag}.grad_s2": g[:, :, ::2, ::2].numpy(),
                         f"{tag}.grad_sum": g.double().sum(dim=(0, 2, 3)).numpy(), f"{tag}.grad_absmax": float(g.abs().max()),
                         f"{tag}.image_s8": img.detach()[:, :, ::8, ::8].numpy(), f"{tag}.clamped_share": frac})
    np.savez_compressed(os.path.join(OUT, "sr_backward.npz"), seed=41, **data)
    print("  wrote sr_backward.npz")


RESIZE_BWD_CASES = [(1, 3, 64, 64, 128, 128, True), (2, 4, 37, 41, 64, 24, True), (1, 8, 96, 96, 24, 24, True), (1, 3, 50, 50, 128, 128, False),
                    (2, 5, 128, 96, 64, 40, False), (1, 2, 512, 512, 128, 128, True)]


def gen_resize_backward():
    """Input gradient of F.interpolate(mode='bilinear', align_corners=False, antialias=...) - the reference's pre-resize of the SR
    head's inputs (superresolution.py:283-286) - by autograd: up- and down-scaling, odd sizes, both antialias settings.  The test
    regenerates x and the cotangent from RandomState(300 + case)."""
    data = {}
    for i, (N, C, H, W, OH, OW, aa) in enumerate(RESIZE_BWD_CASES):
        rng = np.random.RandomState(300 + i)
        x = t(rng.randn(N, C, H, W)).requires_grad_(True)
        cot = t(rng.randn(N, C, OH, OW))
        with torch.enable_grad():
            y = torch.nn.functional.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=False, antialias=aa)
            (g,) = torch.autograd.grad((y * cot).sum(), x)
        stride = 4 if H >= 512 else 1
        data[f"case{i}.grad"] = g.numpy()[:, :, ::stride, ::stride]
        data[f"case{i}.grad_sum"] = g.double().sum(dim=(0, 2, 3)).numpy()
    np.savez_compressed(os.path.join(OUT, "resize_backward.npz"), cases=np.array(RESIZE_BWD_CASES, dtype=np.int64), **data)
    print("  wrote resize_backward.npz")


def gen_sr_backward_r64():
    """gen_sr_backward at neural_rendering_resolution 64 (BASELINE config 1): the 64^2 feature image goes through the head's
    antialiased bilinear pre-resize to 128^2 (superresolution.py:283-286), so the input gradient includes that resize's adjoint."""
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True,
                                   channel_base=32768, channel_max=512, fused_modconv_default="inference_only")
    load(sr, sr_params(41))
    rng = np.random.RandomState(78)
    with torch.enable_grad():
        x = t(rng.randn(1, 32, 64, 64) * 0.5).requires_grad_(True)
        ws = t(rng.randn(1, 14, 512))
        cot = t(rng.randn(1, 3, 512, 512))
        img = sr(x[:, :3], x, ws, noise_mode="none")
        (g,) = torch.autograd.grad((img * cot).sum(), x)
    print(f"    sr_backward_r64: |image| max {float(img.abs().max()):.3g}, |grad| max {float(g.abs().max()):.3g}")
    # inputs are regenerated by the test from RandomState(78): x (x 0.5), ws, cot
    np.savez_compressed(os.path.join(OUT, "sr_backward_r64.npz"), seed=41, grad=g.numpy(), grad_sum=g.double().sum(dim=(0, 2, 3)).numpy(),
                        grad_absmax=float(g.abs().max()), image_s8=img.detach()[:, :, ::8, ::8].numpy())
    print("  wrote sr_backward_r64.npz")


def gen_block_backward():
    """One reference SynthesisBlock (skip architecture, up-sampling conv0 + conv1 + ToRGB + upsample2d skip; conv_clamp 256 as the
    SR head's) under autograd: d<cot_x, x_out> + <cot_img, img_out> / d (x_in, img_in).  Small enough (32 -> 64 channels, 32^2 ->
    64^2) to keep, besides the gradients, the SIGN of every leaky-ReLU unit and the clamp masks of the forward: a unit whose
    pre-activation sits within fp32 rounding of zero may land on the other slope in another implementation of the same forward,
    and the test pins the slopes to the reference's so that every gradient entry can be held to the bar."""
    from training.networks_stylegan2 import SynthesisBlock
    from oracle.dense_params import block_params
    blk = SynthesisBlock(32, 64, w_dim=512, resolution=64, img_channels=3, is_last=False, architecture="skip", conv_clamp=256,
                         use_fp16=False, fused_modconv_default="inference_only").eval().requires_grad_(False)
    load(blk, block_params(91, 32, 64, 512, 64, 3))
    rng = np.random.RandomState(92)
    N = 2
    x = t(rng.randn(N, 32, 32, 32) * 150.0).requires_grad_(True)         # large enough to push some units into the +-256 clamp
    img = t(rng.randn(N, 3, 32, 32)).requires_grad_(True)
    ws = t(rng.randn(N, 3, 512))
    cot_x, cot_img = t(rng.randn(N, 64, 64, 64)), t(rng.randn(N, 3, 64, 64))
    acts = {}
    hooks = [getattr(blk, n).register_forward_hook(lambda m, i, o, n=n: acts.__setitem__(n, o.detach())) for n in ("conv0", "conv1", "torgb")]
    with torch.enable_grad():
        xo, io = blk(x, img, ws, noise_mode="const")
        gx, gi = torch.autograd.grad((xo * cot_x).sum() + (io * cot_img).sum(), (x, img))
    for h in hooks:
        h.remove()
    neg0, neg1 = (acts["conv0"] < 0).numpy(), (acts["conv1"] < 0).numpy()
    cl0, cl1, cly = (acts["conv0"].abs() >= 256).numpy(), (acts["conv1"].abs() >= 256).numpy(), (acts["torgb"].abs() >= 256).numpy()
    print(f"    block_backward: clamped units conv0 {cl0.mean():.4f} conv1 {cl1.mean():.4f} torgb {cly.mean():.4f}; |gx| max {float(gx.abs().max()):.3g}")
    # inputs and cotangents are regenerated by the test from RandomState(92) in this order: x (x 150), img, ws, cot_x, cot_img
    np.savez_compressed(os.path.join(OUT, "block_backward.npz"), seed=91, x_out=xo.detach().numpy()[:, :, ::8, ::8], img_out=io.detach().numpy()[:, :, ::4, ::4],
                        grad_x=gx.numpy(), grad_img=gi.numpy(), neg0=np.packbits(neg0), neg1=np.packbits(neg1), clamp0=np.packbits(cl0),
                        clamp1=np.packbits(cl1), clampy=np.packbits(cly), torch_version=np.array(torch.__version__))
    print("  wrote block_backward.npz")


E2E_KW = dict(
    rendering_kwargs=dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True,
                          c_gen_conditioning_zero=False, c_scale=1, superresolution_noise_mode="none", depth_resolution=12,
                          depth_resolution_importance=12, ray_start=2.25, ray_end=3.3, box_warp=1,
                          disparity_space_sampling=False, clamp_mode="softplus", decoder_lr_mul=1),
    channel_base=4096, channel_max=32)


def gen_e2e():
    """Whole TriPlaneGenerator forward (mapping -> backbone -> normalise -> render -> SR) on a reduced-width
    backbone, 32^2 neural render, 12+12 samples, injected jitter, plus an appearance-swapped second call
    and a sample_mixed() point query."""
    import math
    from camera_utils import FOV_to_intrinsics, LookAtPoseSampler
    from training.triplane import TriPlaneGenerator
    from oracle import e2e_oracle
    from oracle.dense_params import generator_params
    from oracle.gen_golden import InjectRand
    rk = dict(E2E_KW["rendering_kwargs"])
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=E2E_KW["channel_base"], channel_max=E2E_KW["channel_max"],
                          fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    p = generator_params(51, E2E_KW["channel_base"], E2E_KW["channel_max"])
    load(G, p)
    rng = np.random.RandomState(52)
    N, R, D, Di = 2, 32, 12, 12
    z = t(rng.randn(N, 512))
    c2w = torch.cat([LookAtPoseSampler.sample(math.pi / 2 + y, math.pi / 2 - 0.2, torch.tensor([0, 0, 0.2]), radius=2.7) for y in (0.4, -0.3)], 0)
    c = torch.cat([c2w.reshape(N, 16), FOV_to_intrinsics(18.837).reshape(1, 9).repeat(N, 1)], 1)
    u_c = rng.rand(N, R * R, D).astype(np.float32)
    u_f = rng.rand(N * R * R, Di).astype(np.float32)
    ws = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=14)
    check("e2e.mapping", e2e_oracle.mapping(p, z, c, rk, 0.7, 14), ws, 1e-4)
    data = dict(z=z.numpy(), c=c.numpy(), ws=ws.numpy(), u_coarse=u_c, u_fine=u_f, seed=51, R=R)
    for tag, kw in dict(plain={}, swap=dict(planes_mean=1, planes_var=0)).items():
        with InjectRand([u_c, u_f]):
            ref = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const", **kw)
        mine = e2e_oracle.synthesis(p, ws, c, rk, R, u_c, u_f, noise_mode="const", **kw)
        for k in ("image", "image_seg", "image_raw", "image_depth", "plane_mean", "plane_var"):
            check(f"e2e.{tag}.{k}", torch.from_numpy(np.ascontiguousarray(mine[k])), ref[k], 3e-4)
        data.update({f"{tag}.image_s4": ref["image"][:, :, ::4, ::4].numpy(), f"{tag}.image_mean": float(ref["image"].mean()),
                     f"{tag}.image_seg": ref["image_seg"].numpy(), f"{tag}.image_raw": ref["image_raw"].numpy(),
                     f"{tag}.image_depth": ref["image_depth"].numpy(), f"{tag}.plane_mean": ref["plane_mean"].numpy(),
                     f"{tag}.plane_var": ref["plane_var"].numpy()})
    coords = t((rng.rand(N, 300, 3) - 0.5) * 1.1)
    ref = G.sample_mixed(coords, None, ws, noise_mode="const")
    mine = e2e_oracle.sample_mixed(p, coords.numpy(), ws, rk)
    for k in ("rgb", "sigma", "seg"):
        check("e2e.sample." + k, torch.from_numpy(mine[k]), ref[k], 3e-4)
        data["sample." + k] = ref[k].numpy()
    data["sample.coords"] = coords.numpy()
    np.savez_compressed(os.path.join(OUT, "dense_e2e.npz"), **data)
    print("  wrote dense_e2e.npz")


def gen_sr_variants():
    """The other super-resolution heads (superresolution.py:29-155): 8X, 4X, 2X, Deepfp32.  Reference outputs only."""
    from training import superresolution as ref_sr
    from oracle.dense_params import params_by_name
    data = {}
    for name, res, in_res, kw in (("SuperresolutionHybrid8X", 512, 64, dict(sr_antialias=True)), ("SuperresolutionHybrid4X", 256, 64, dict(sr_antialias=True)),
                                  ("SuperresolutionHybrid4X", 256, 128, dict(sr_antialias=False)), ("SuperresolutionHybrid2X", 128, 96, dict(sr_antialias=True)),
                                  ("SuperresolutionHybridDeepfp32", 256, 64, dict())):
        tag = f"{name}.{in_res}"
        net = getattr(ref_sr, name)(channels=32, img_resolution=res, sr_num_fp16_res=4, **kw)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        load(net, params_by_name(81, shapes))
        rng = np.random.RandomState(zlib_crc(tag))
        x = t(rng.randn(1, 32, in_res, in_res)); ws = t(rng.randn(1, 14, 512))
        out = net(x[:, :3].contiguous(), x, ws, noise_mode="const")
        print(f"    reference {tag:36s} out {tuple(out.shape)} |max| {float(out.abs().max()):.3g}")
        data.update({tag + ".out_s4": out[:, :, 1::4, 3::4].numpy(),       # x, ws: regenerated from crc32(tag) by the test
                     tag + ".out_mean": float(out.double().mean()), tag + ".keys": np.array(sorted(shapes))})
    np.savez_compressed(os.path.join(OUT, "dense_sr_variants.npz"), seed=81, **data)
    print("  wrote dense_sr_variants.npz")


SR_BWD_VARIANTS = (("SuperresolutionHybrid8X", 512, 64, dict(sr_antialias=True)), ("SuperresolutionHybrid4X", 256, 64, dict(sr_antialias=True)),
                   ("SuperresolutionHybrid4X", 256, 128, dict(sr_antialias=False)), ("SuperresolutionHybrid2X", 128, 96, dict(sr_antialias=True)),
                   ("SuperresolutionHybridDeepfp32", 256, 128, dict()))


def gen_sr_backward_variants():
    """Input gradient of the other super-resolution heads by the reference's autograd: heads whose first block does not up-sample
    (4X, 2X, Deepfp32: SynthesisBlockNoUp), fp32 heads without clamp, inputs smaller / larger than the head's input resolution
    (4X at its own 128^2: no resize)."""
    from training import superresolution as ref_sr
    from oracle.dense_params import params_by_name
    data = {}
    for name, res, in_res, kw in SR_BWD_VARIANTS:
        tag = f"{name}.{in_res}"
        net = getattr(ref_sr, name)(channels=32, img_resolution=res, sr_num_fp16_res=4, **kw)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        load(net, params_by_name(81, shapes))
        rng = np.random.RandomState(zlib_crc("bwd." + tag))
        with torch.enable_grad():
            x = t(rng.randn(1, 32, in_res, in_res) * 0.5).requires_grad_(True)
            ws = t(rng.randn(1, 14, 512))
            out = net(x[:, :3].clone(), x, ws, noise_mode="none")      # clone: SynthesisBlockNoUp adds to its image in place (:250)
            cot = t(rng.randn(*out.shape))
            (g,) = torch.autograd.grad((out * cot).sum(), x)
        print(f"    reference {tag:36s} out {tuple(out.shape)} |grad| max {float(g.abs().max()):.3g}")
        # x (x 0.5), ws, cot: regenerated from crc32("bwd." + tag) by the test, in this order
        st = 4 if in_res >= 128 else 2                       # kept entries: every st-th pixel (the file stays small)
        data.update({tag + ".stride": st, tag + ".grad_s": g[:, :, ::st, ::st].numpy(), tag + ".grad_sum": g.double().sum(dim=(0, 2, 3)).numpy(),
                     tag + ".grad_absmax": float(g.abs().max()), tag + ".out_s8": out.detach()[:, :, ::8, ::8].numpy()})
    np.savez_compressed(os.path.join(OUT, "sr_backward_variants.npz"), seed=81, **data)
    print("  wrote sr_backward_variants.npz")


def zlib_crc(s):
    import zlib
    return zlib.crc32(s.encode()) & 0x7FFFFFFF


def gen_e2e_full():
    """The FFHQ-size generator (channel_base 32768, channel_max 512: 30.7 M parameters) end to end, one view, 64^2
    neural render, 24+24 samples, injected jitter.  Reference outputs only (sub-sampled where large)."""
    import math
    from camera_utils import FOV_to_intrinsics, LookAtPoseSampler
    from training.triplane import TriPlaneGenerator
    from oracle.dense_params import generator_params
    from oracle.gen_golden import InjectRand
    rk = dict(E2E_KW["rendering_kwargs"], depth_resolution=24, depth_resolution_importance=24)
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=32768, channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    p = generator_params(71, 32768, 512)
    load(G, p)
    assert sum(v.numel() for v in G.parameters()) == 30665223
    rng = np.random.RandomState(72)
    N, R, D, Di = 1, 64, 24, 24
    z = t(rng.randn(N, 512))
    c2w = LookAtPoseSampler.sample(math.pi / 2 + 0.25, math.pi / 2 - 0.15, torch.tensor([0, 0, 0.2]), radius=2.7)
    c = torch.cat([c2w.reshape(N, 16), FOV_to_intrinsics(18.837).reshape(1, 9)], 1)
    u_c = rng.rand(N, R * R, D).astype(np.float32)
    u_f = rng.rand(N * R * R, Di).astype(np.float32)
    with InjectRand([u_c, u_f]):
        ref = G(z, c, truncation_psi=0.7, truncation_cutoff=14, neural_rendering_resolution=R, noise_mode="const")
    data = dict(z=z.numpy(), c=c.numpy(), u_coarse=u_c, u_fine=u_f, seed=71, R=R, D=D, Di=Di,
                image_s4=ref["image"][:, :, 1::4, 2::4].numpy(), image_mean=float(ref["image"].mean()),
                image_seg=ref["image_seg"].numpy(), image_raw=ref["image_raw"].numpy(), image_depth=ref["image_depth"].numpy(),
                plane_mean=ref["plane_mean"].numpy(), plane_var=ref["plane_var"].numpy())
    for k in ("image", "image_seg", "image_raw", "image_depth"):
        print(f"    reference {k:12s} |max| {float(ref[k].abs().max()):.3g}")
    np.savez_compressed(os.path.join(OUT, "dense_e2e_full.npz"), **data)
    print("  wrote dense_e2e_full.npz")


def gen_density_grid():
    """gen_samples.py shape extraction (:79-101 create_samples, :186-214 the sigma sweep + flip + border trim) on the reduced
    generator of gen_e2e: N=16 grid.  create_samples is the reference's own function (gen_samples.py imports `mrcfile`, absent
    here and used only when a .mrc file is written: an empty placeholder module lets the import proceed); the sweep calls the
    reference G.sample_mixed in chunks like gen_samples.py:197-201; flip + trim restate :205-216."""
    import types
    sys.modules.setdefault("mrcfile", types.ModuleType("mrcfile"))
    import gen_samples as ref_gs
    from training.triplane import TriPlaneGenerator
    from oracle.dense_params import generator_params
    rk = dict(E2E_KW["rendering_kwargs"])
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=E2E_KW["channel_base"], channel_max=E2E_KW["channel_max"],
                          fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    load(G, generator_params(51, E2E_KW["channel_base"], E2E_KW["channel_max"]))
    e2e = np.load(os.path.join(OUT, "dense_e2e.npz"))
    ws = t(e2e["ws"])[:1]
    shape_res, max_batch = 16, 1500
    samples, voxel_origin, voxel_size = ref_gs.create_samples(N=shape_res, voxel_origin=[0, 0, 0], cube_length=rk["box_warp"] * 1)
    sigmas = torch.zeros((samples.shape[0], samples.shape[1], 1))
    head = 0
    while head < samples.shape[1]:
        sigmas[:, head:head + max_batch] = G.sample_mixed(samples[:, head:head + max_batch], None, ws, noise_mode="const")["sigma"]
        head += max_batch
    grid = sigmas.reshape((shape_res, shape_res, shape_res)).numpy()
    vol = np.flip(grid, 0).copy()
    pad = int(30 * shape_res / 256)
    vol[:pad] = vol[-pad:] = -1000
    vol[:, :pad] = vol[:, -pad:] = -1000
    vol[:, :, :pad] = vol[:, :, -pad:] = -1000
    print(f"    reference sigma grid |max| {np.abs(grid).max():.3g}, voxel_size {voxel_size:.6f}, pad {pad}")
    np.savez_compressed(os.path.join(OUT, "density_grid.npz"), samples=samples.numpy(), voxel_origin=np.asarray(voxel_origin, np.float64),
                        voxel_size=float(voxel_size), shape_res=shape_res, max_batch=max_batch, sigma_grid=grid, sigma_volume=vol, pad=pad)
    print("  wrote density_grid.npz")


def _full_generator(rk):
    from training.triplane import TriPlaneGenerator
    from oracle.dense_params import generator_params
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=32768, channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
    load(G, generator_params(71, 32768, 512))
    assert sum(v.numel() for v in G.parameters()) == 30665223
    return G


def gen_e2e_cfg1():
    """BASELINE config 1 exactly: full-size generator forward(), N=1, 64^2 neural render, 48 + 48
    samples, z = RandomState(0).randn(1,512), c = LookAtPoseSampler.sample(pi/2, pi/2, [0,0,0.2], radius=2.7) + FOV 18.837,
    truncation_psi 1, noise_mode 'const'.  Jitter is regenerated from `u_seed` by the test."""
    import math
    from camera_utils import FOV_to_intrinsics, LookAtPoseSampler
    from oracle.gen_golden import InjectRand
    R, D, Di, u_seed = 64, 48, 48, 73
    G = _full_generator(dict(E2E_KW["rendering_kwargs"], depth_resolution=D, depth_resolution_importance=Di))
    z = t(np.random.RandomState(0).randn(1, 512))
    c2w = LookAtPoseSampler.sample(math.pi / 2, math.pi / 2, torch.tensor([0, 0, 0.2]), radius=2.7)
    c = torch.cat([c2w.reshape(1, 16), FOV_to_intrinsics(18.837).reshape(1, 9)], 1)
    rng = np.random.RandomState(u_seed)
    u_c = rng.rand(1, R * R, D).astype(np.float32)
    u_f = rng.rand(R * R, Di).astype(np.float32)
    with InjectRand([u_c, u_f]):
        ref = G(z, c, neural_rendering_resolution=R, noise_mode="const")
    data = dict(z=z.numpy(), c=c.numpy(), u_seed=u_seed, seed=71, R=R, D=D, Di=Di,
                image_s4=ref["image"][:, :, 1::4, 2::4].numpy(), image_mean=float(ref["image"].double().mean()),
                image_seg=ref["image_seg"].numpy(), image_raw=ref["image_raw"].numpy(), image_depth=ref["image_depth"].numpy(),
                plane_mean=ref["plane_mean"].numpy(), plane_var=ref["plane_var"].numpy(), torch_version=np.array(torch.__version__))
    for k in ("image", "image_seg", "image_raw", "image_depth"):
        print(f"    reference {k:12s} |max| {float(ref[k].abs().max()):.3g}")
    np.savez_compressed(os.path.join(OUT, "dense_e2e_cfg1.npz"), **data)
    print("  wrote dense_e2e_cfg1.npz")


def gen_e2e_cfg3():
    """BASELINE config 3's data path at full size: synthesis() of the FFHQ-size generator at neural_rendering_resolution
    512 with 64 samples (single pass), so the 32-channel 512^2 feature image goes through the antialiased 512 -> 128
    down-resize into the SR head (superresolution.py:279-290).  One view (the reference materialises ~6.4 GB of gather
    results per plane set at this size).  Jitter regenerated from `u_seed`; outputs stored strided + fp64 means."""
    import math
    from camera_utils import FOV_to_intrinsics, LookAtPoseSampler
    from oracle.gen_golden import InjectRand
    R, D, u_seed = 512, 64, 75
    G = _full_generator(dict(E2E_KW["rendering_kwargs"], depth_resolution=D, depth_resolution_importance=0))
    rng = np.random.RandomState(74)
    z = t(rng.randn(1, 512))
    c2w = LookAtPoseSampler.sample(math.pi / 2 - 0.3, math.pi / 2 - 0.1, torch.tensor([0, 0, 0.2]), radius=2.7)
    c = torch.cat([c2w.reshape(1, 16), FOV_to_intrinsics(18.837).reshape(1, 9)], 1)
    u_c = np.random.RandomState(u_seed).rand(1, R * R, D).astype(np.float32)
    ws = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=14)
    with InjectRand([u_c]):
        ref = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const")
    data = dict(z=z.numpy(), c=c.numpy(), ws=ws.numpy(), u_seed=u_seed, seed=71, R=R, D=D, Di=0,
                image_s4=ref["image"][:, :, 1::4, 2::4].numpy(), image_raw_s3=ref["image_raw"][:, :, 1::3, 2::3].numpy(),
                image_seg_s4=ref["image_seg"][:, :, 2::4, 1::4].numpy(), image_depth_s2=ref["image_depth"][:, :, ::2, 1::2].numpy(),
                plane_mean=ref["plane_mean"].numpy(), plane_var=ref["plane_var"].numpy(), torch_version=np.array(torch.__version__))
    for k in ("image", "image_seg", "image_raw", "image_depth"):
        data[k + "_mean"] = ref[k].double().mean(dim=(0, 2, 3)).numpy()
        data[k + "_absmax"] = float(ref[k].abs().max())
        print(f"    reference {k:12s} |max| {float(ref[k].abs().max()):.3g}")
    np.savez_compressed(os.path.join(OUT, "dense_e2e_cfg3.npz"), **data)
    print("  wrote dense_e2e_cfg3.npz")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else None         # e.g. `python oracle/gen_golden_dense.py synthesis_full`
    if only:
        globals()["gen_" + only]()
        sys.exit(0)
    gen_mapping()
    gen_layers()
    gen_synthesis()
    gen_synthesis_full()
    gen_sr()
    gen_sr_backward()
    gen_sr_backward_r64()
    gen_sr_backward_variants()
    gen_resize_backward()
    gen_block_backward()
    gen_e2e()
    gen_e2e_full()
    gen_e2e_cfg1()
    gen_e2e_cfg3()
    gen_density_grid()
    gen_sr_variants()
    for f in sorted(os.listdir(OUT)):
        if f.startswith("dense_"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
