"""GPU parity of the dense half (mapping, modulated convs, ToRGB+skip, resize, backbone, SR head) through
the C ABI, against the golden vectors captured from the reference and the torch-CPU oracle.

Tolerances: 1e-3 max-abs is the north_star bar for the whole path in fp32-grade math ('bf16x3': fp32
operands split into bf16 hi+lo, fp32 accumulate).  The plain-bf16 throughput mode ('bf16', BASELINE
config 3) is held to a relative bound stated at each test.
"""
import numpy as np
import pytest
import torch

from oracle import dense_oracle as dor
from oracle.dense_params import mapping_params, sr_params, synthesis_params
from tests._golden import load
from tests.test_dense_oracle_golden import LAYERS, layer_case

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def t(a, dev=None):
    x = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return x.to(dev) if dev is not None else x


def maxerr(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def load_module(m, params, dev):
    sd = m.state_dict()
    for k, v in params.items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = v.clone()
    m.load_state_dict(sd)
    return m.to(dev).eval()


def test_layout_roundtrip_and_stats(dev):
    from nerffaceediting_amd import dense_ops, ops
    rng = np.random.RandomState(0)
    x = t(rng.randn(2, 96, 9, 7) * 2 + 0.3, dev)
    y = dense_ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(dense_ops.nhwc_to_nchw(y), x)
    assert torch.equal(dense_ops.nhwc_to_planes(y), ops.plane_pack(x))
    m0, s0 = ops.plane_stats(x)
    m1, s1 = dense_ops.plane_stats_nhwc(y)
    assert maxerr(m1, m0.cpu()) <= 1e-6 and maxerr(s1, s0.cpu()) <= 1e-6
    for C in (1, 3, 4, 15, 33):               # 1: copy, 3 / 15: the per-pixel kernel of the image / segmentation outputs, others: LDS tiles
        xs = t(rng.randn(2, C, 37, 19), dev)
        assert torch.equal(dense_ops.nhwc_to_nchw(xs.permute(0, 2, 3, 1).contiguous()), xs), C


def test_mapping_network(dev):
    from nerffaceediting_amd.training.networks_stylegan2 import MappingNetwork
    z = load("dense_mapping")
    p = mapping_params(int(z["seed"]), int(z["z_dim"]), int(z["c_dim"]), int(z["w_dim"]))
    m = load_module(MappingNetwork(int(z["z_dim"]), int(z["c_dim"]), int(z["w_dim"]), int(z["num_ws"]), num_layers=2), p, dev)
    for tag, (psi, cut) in dict(a=(1.0, None), b=(0.7, None), c=(0.5, 4)).items():
        ws = m(t(z["z"], dev), t(z["c"], dev), truncation_psi=psi, truncation_cutoff=cut)
        assert maxerr(ws, z["ws." + tag]) <= 2e-5, tag


@pytest.mark.parametrize("math", ["bf16x3", "bf16", "fp16"])
@pytest.mark.parametrize("tag", LAYERS)
def test_modulated_conv_layers(tag, math, dev):
    from nerffaceediting_amd import dense_ops
    from nerffaceediting_amd.training.networks_stylegan2 import SynthesisLayer, ToRGBLayer
    z = load("dense_layers")
    cfg, p = layer_case(z, tag)
    w_dim = int(z["w_dim"])
    x, w = dense_ops.nchw_to_nhwc(t(z[tag + ".x"], dev)), t(z[tag + ".w"], dev)
    if cfg["torgb"]:
        m = load_module(ToRGBLayer(cfg["cin"], cfg["cout"], w_dim, conv_clamp=cfg["clamp"]), p, dev)
        y = m.forward_nhwc(x, w, conv_math=math)
    else:
        m = load_module(SynthesisLayer(cfg["cin"], cfg["cout"], w_dim, cfg["res"], up=cfg["up"], conv_clamp=cfg["clamp"]), p, dev)
        y = m.forward_nhwc(x, w, noise_mode="const", gain=cfg["gain"], conv_math=math)
    ref = z[tag + ".out"]
    e = maxerr(dense_ops.nhwc_to_nchw(y), ref)
    print(tag, math, e, float(np.abs(ref).max()))
    # bf16x3: fp32-grade.  bf16: 8-bit mantissa operands, K up to 432 terms -> ~1e-2 of the output scale.  fp16 (round 4): 11-bit
    # operands, 8 x finer
    assert e <= {"bf16x3": 1e-4, "bf16": 3e-2 * float(np.abs(ref).max()), "fp16": 4e-3 * float(np.abs(ref).max())}[math]


def test_torgb_skip_matches_upsample2d(dev):
    """img = upsample2d(img) + torgb(x)  (networks_stylegan2.py:450-457), fused in the ToRGB epilogue."""
    from nerffaceediting_amd import dense_ops
    from nerffaceediting_amd.training.networks_stylegan2 import ToRGBLayer
    z = load("dense_layers")
    cfg, p = layer_case(z, "torgb96")
    m = load_module(ToRGBLayer(cfg["cin"], cfg["cout"], int(z["w_dim"])), p, dev)
    rng = np.random.RandomState(3)
    prev = t(rng.randn(2, 96, cfg["res"] // 2, cfg["res"] // 2))
    x, w = t(z["torgb96.x"]), t(z["torgb96.w"])
    want = dor.upsample2d(prev) + dor.torgb_layer(p, x, w)
    got = m.forward_nhwc(dense_ops.nchw_to_nhwc(x.to(dev)), w.to(dev), skip=dense_ops.nchw_to_nhwc(prev.to(dev)))
    assert maxerr(dense_ops.nhwc_to_nchw(got), want) <= 1e-4
    planes = m.forward_nhwc(dense_ops.nchw_to_nhwc(x.to(dev)), w.to(dev), skip=dense_ops.nchw_to_nhwc(prev.to(dev)), out_planes=True)
    assert torch.equal(planes, dense_ops.nhwc_to_planes(got))


def test_resize_bilinear(dev):
    from nerffaceediting_amd import dense_ops
    z = load("dense_layers")
    for tag in ("down_aa", "up_aa", "down_noaa", "odd_aa"):
        out = z[f"resize.{tag}.out"]
        y = dense_ops.resize_bilinear(dense_ops.nchw_to_nhwc(t(z[f"resize.{tag}.x"], dev)), out.shape[2], out.shape[3],
                                      bool(int(z[f"resize.{tag}.aa"])))
        assert maxerr(dense_ops.nhwc_to_nchw(y), out) <= 2e-6, tag
    # every (direction, antialias) combination against torch's own kernel on the device (the far edge of a plain bilinear
    # up-sampling clamps its second tap onto the first: SuperresolutionHybridDeepfp32, superresolution.py:146-149)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(2, 40, 56, 8, generator=g).to(dev)
    for aa in (True, False):
        for oh, ow in ((80, 112), (128, 128), (20, 33), (40, 56), (41, 57)):
            y = dense_ops.resize_bilinear(x, oh, ow, aa)
            ref = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear", align_corners=False, antialias=aa)
            assert float((y - ref.permute(0, 2, 3, 1)).abs().max()) <= 1e-5, (aa, oh, ow)
    # the SR head's own case (superresolution.py:283-286): 32- and 3-channel images, 4x antialiased down-scale (9 taps per axis: the
    # 4-channel fast kernel and the scalar one), and a 6x down-scale whose 13 taps exceed the fast kernel's table
    for C, (H, oh) in ((32, (64, 16)), (3, (64, 16)), (8, (96, 16))):
        x = torch.randn(2, H, H, C, generator=g).to(dev)
        y = dense_ops.resize_bilinear(x, oh, oh, True)
        ref = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(oh, oh), mode="bilinear", align_corners=False, antialias=True)
        assert float((y - ref.permute(0, 2, 3, 1)).abs().max()) <= 1e-5, (C, H, oh)


def test_reduced_synthesis_network(dev):
    from nerffaceediting_amd.training.networks_stylegan2 import SynthesisNetwork
    z = load("dense_synthesis")
    p = synthesis_params(int(z["seed"]), int(z["w_dim"]), int(z["res"]), 96, int(z["channel_base"]), int(z["channel_max"]))
    net = load_module(SynthesisNetwork(int(z["w_dim"]), int(z["res"]), 96, channel_base=int(z["channel_base"]),
                                       channel_max=int(z["channel_max"]), num_fp16_res=0, conv_clamp=None), p, dev)
    out = net(t(z["ws"], dev), noise_mode="const")
    e = maxerr(out, z["out"])
    print("synthesis 32px", e)
    assert e <= TOL
    planes = net.forward_nhwc(t(z["ws"], dev), out_planes=True, noise_mode="const")
    from nerffaceediting_amd import ops
    assert torch.equal(planes, ops.plane_pack(out))


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
def test_full_width_synthesis_network(math, dev):
    """FFHQ-width backbone (512 channels, 256 px: split-K small layers, LDS-DMA conv path, chained up-sampling layers)
    against the sub-sampled output of the reference SynthesisNetwork."""
    from nerffaceediting_amd.training.networks_stylegan2 import SynthesisNetwork
    z = load("dense_synthesis_full")
    p = synthesis_params(int(z["seed"]), int(z["w_dim"]), int(z["res"]), 96, int(z["channel_base"]), int(z["channel_max"]))
    net = load_module(SynthesisNetwork(int(z["w_dim"]), int(z["res"]), 96, channel_base=int(z["channel_base"]),
                                       channel_max=int(z["channel_max"]), num_fp16_res=0, conv_clamp=None), p, dev)
    net.conv_math = math
    out = net(t(z["ws"], dev), noise_mode="const")
    scale = float(z["absmax"])
    e = maxerr(out[:, :, 3::8, 5::8], z["out_s8"])
    print("full-width synthesis", math, e, "of", scale)
    assert e <= (1e-4 if math == "bf16x3" else 3e-2) * scale
    assert maxerr(out.mean(dim=(2, 3)), z["ch_mean"]) <= (1e-4 if math == "bf16x3" else 5e-2)


@pytest.mark.parametrize("tag", ["r64", "r128"])
def test_superresolution_8xdc(tag, dev):
    from nerffaceediting_amd.training.superresolution import SuperresolutionHybrid8XDC
    z = load("dense_sr")
    sr = load_module(SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True), sr_params(int(z["seed"])), dev)
    x = t(z[tag + ".x"], dev)
    y = sr(x[:, :3].contiguous(), x, t(z[tag + ".ws"], dev), noise_mode="none")
    assert y.shape == (1, 3, 512, 512)
    e = maxerr(y[:, :, ::4, ::4], z[tag + ".out_s4"])
    print("sr", tag, e)
    assert e <= TOL
    assert abs(float(y.mean()) - float(z[tag + ".out_mean"])) <= 1e-4


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
@pytest.mark.parametrize("shape", [(2, 40, 72, 48, 128), (1, 8, 32, 16, 64), (3, 33, 95, 64, 64), (2, 8, 8, 256, 64), (1, 16, 16, 512, 96), (3, 4, 4, 128, 32), (4, 250, 255, 32, 128)])
def test_conv3x3_fast_path_matches_generic(shape, math, dev):
    """The LDS-DMA 3x3 path (pre-split activations) against the generic kernel on ragged tiles, and against a
    torch fp32 conv of the same modulated/demodulated layer (modulated_conv2d, networks_stylegan2.py:34-91)."""
    from nerffaceediting_amd import _lib, dense_ops as D
    N, H, W, cin, cout = shape
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(N, H, W, cin, generator=g).to(dev)
    styles = (torch.randn(N, cin, generator=g) * 0.5 + 1.0).to(dev)
    weight = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    noise = torch.randn(H, W, generator=g).to(dev)
    packed, wsq = D.conv_pack(weight)
    dcoef = D.conv_demod(styles, wsq)
    kw = dict(bias=bias, dcoef=dcoef, noise=noise, noise_strength=0.3, lrelu=True, act_gain=2 ** 0.5, clamp=256.0, math=math)
    assert _lib.load().nfe_conv_scratch_floats(_lib.NFE_CONV_3X3, D.MATH[math], N, H, W, cin, cout) > 0
    fast = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3, **kw)
    D.FAST_PATH = False
    try:
        slow = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3, **kw)
    finally:
        D.FAST_PATH = True
    assert float((fast - slow).abs().max()) <= 1e-5 * float(slow.abs().max())
    w = weight[None] * styles[:, None, :, None, None]
    w = w * torch.rsqrt((w * w).sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
    ref = torch.cat([torch.nn.functional.conv2d(x[i:i + 1].permute(0, 3, 1, 2), w[i], padding=1) for i in range(N)], 0)
    ref = ref + noise[None, None] * 0.3 + bias[None, :, None, None]
    ref = (torch.nn.functional.leaky_relu(ref, 0.2) * 2 ** 0.5).clamp(-256, 256).permute(0, 2, 3, 1)
    tol = (2e-5 if math == "bf16x3" else 2e-2) * float(ref.abs().max())
    assert float((fast - ref).abs().max()) <= tol


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
@pytest.mark.parametrize("shape", [(2, 40, 72, 48, 96), (1, 8, 32, 16, 32), (2, 33, 63, 32, 64), (2, 4, 4, 256, 64), (1, 16, 16, 128, 32)])
def test_upconv_fast_path_matches_generic(shape, math, dev):
    """Up-sampling layer (transposed conv + FIR, conv2d_resample.py:114-128) through the LDS-DMA path vs the generic kernel."""
    from nerffaceediting_amd import _lib, dense_ops as D
    N, H, W, cin, cout = shape
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(N, H, W, cin, generator=g).to(dev)
    styles = (torch.randn(N, cin, generator=g) * 0.5 + 1.0).to(dev)
    weight = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    noise = torch.randn(2 * H, 2 * W, generator=g).to(dev)
    packed, wsq = D.conv_pack(weight)
    dcoef = D.conv_demod(styles, wsq)
    kw = dict(bias=bias, dcoef=dcoef, noise=noise, noise_strength=0.3, lrelu=True, act_gain=2 ** 0.5, clamp=None, math=math)
    need = _lib.load().nfe_conv_scratch_floats(_lib.NFE_CONV_3X3_UP2, D.MATH[math], N, H, W, cin, cout)
    assert need > N * (2 * H + 1) * (2 * W + 1) * cout
    fast = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3_UP2, **kw)
    D.FAST_PATH = False
    try:
        slow = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3_UP2, **kw)
    finally:
        D.FAST_PATH = True
    assert fast.shape == (N, 2 * H, 2 * W, cout)
    assert float((fast - slow).abs().max()) <= 1e-5 * float(slow.abs().max())


_UP_FUSED_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nerffaceediting_amd import _lib, dense_ops as D
dev = torch.device("cuda:0")
out = {}
for ci, (N, H, W, cin, cout) in enumerate([(2, 40, 72, 32, 64), (1, 64, 64, 32, 96), (2, 33, 63, 16, 32), (1, 128, 128, 32, 64),
                                           (2, 24, 40, 64, 32), (1, 72, 61, 64, 64)]):      # 64 input channels: split-bf16 takes the strip kernel too
    for math in ("bf16x3", "bf16"):
        g = torch.Generator(device="cpu").manual_seed(21 + ci)
        x = torch.randn(N, H, W, cin, generator=g).to(dev)
        styles = (torch.randn(N, cin, generator=g) * 0.5 + 1.0).to(dev)
        nstyles = (torch.randn(N, cout, generator=g) * 0.5 + 1.0).to(dev)
        weight = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
        bias = torch.randn(cout, generator=g).to(dev)
        noise = torch.randn(2 * H, 2 * W, generator=g).to(dev)
        packed, wsq = D.conv_pack(weight)
        dcoef = D.conv_demod(styles, wsq)
        kw = dict(bias=bias, dcoef=dcoef, noise=noise, noise_strength=0.3, lrelu=True, act_gain=2 ** 0.5, clamp=2.5, math=math)
        o = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3_UP2, **kw)
        out[f"out_{ci}_{math}"] = o.cpu().numpy()
        if cout % 16 == 0:
            o2, sp = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3_UP2, next_styles=nstyles, want_out=True, **kw)
            nbytes = N * 2 * H * 2 * W * cout * 2 * (2 if math == "bf16x3" else 1)       # bf16 hi (+ lo) planes; the buffer may be padded
            out[f"split_{ci}_{math}"] = sp.data.cpu().numpy().view(np.uint8).reshape(-1)[:nbytes]
            out[f"out2_{ci}_{math}"] = o2.cpu().numpy()
        out[f"how_{ci}_{math}"] = np.frombuffer(D.describe(_lib.NFE_CONV_3X3_UP2, math, N, H, W, cin, cout).encode(), dtype=np.uint8)
np.savez(sys.argv[2], **out)
"""


def test_fused_up_layer_is_bit_identical_to_scratch_form(dev, tmp_path):
    """the up-sampling layers whose 4x4 FIR runs inside the transposed-conv kernel give the SAME BITS as the (2H+1)^2
    fp32 scratch + upfir_kernel form - fp32 output and the consumer's bf16 image, odd sizes, both math modes, noise, bias, clamp - in
    all three fused forms: the strip kernel (round 6: upconv_strip_kernel + upconv_seam_kernel, several segments per strip and several
    strips per image in these shapes), the overlapping tiles where the selection rule still takes them (split-bf16, <= 32 input
    channels), and the overlapping tiles everywhere (NFE_UP_STRIP=0).  The switches are read once per process, so each form runs in its
    own child process (NFE_UP_FUSED=0 / 1 with the per-mode thresholds lifted)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "up_fused.py"
    script.write_text(_UP_FUSED_SCRIPT)
    res = {}
    lifted = {"NFE_UP_FUSED": "1", "NFE_UP_FUSED_CIN_X3": "512", "NFE_UP_FUSED_CIN_BF16": "512"}
    for tag, env in (("scratch", {"NFE_UP_FUSED": "0"}), ("fused", lifted), ("segs3", dict(lifted, NFE_UP_STRIP_SEGS="3")), ("tiles", dict(lifted, NFE_UP_STRIP="0"))):
        f = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, str(script), root, str(f)], check=True, env=dict(os.environ, **env), timeout=600)
        res[tag] = np.load(f)
    keys = sorted(res["scratch"].files)
    strips = 0
    for tag in ("fused", "segs3", "tiles"):
        assert keys == sorted(res[tag].files) and len(keys) >= 30
        for k in keys:
            a, b = res["scratch"][k], res[tag][k]
            if k.startswith("how_"):
                assert b"upfir" in a.tobytes() and b"fused FIR" in b.tobytes(), (k, a.tobytes(), b.tobytes())
                assert tag != "tiles" or b"overlapping tiles" in b.tobytes()
                strips += b"(strips)" in b.tobytes()
            else:
                assert a.shape == b.shape and np.array_equal(a, b), (tag, k)
    assert strips >= 2 * 8                      # every bf16 shape and the 64-channel split-bf16 shapes ran the strip kernel, in both strip runs


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 512, 96), (1, 32, 32, 256, 3), (3, 4, 4, 128, 96), (2, 64, 64, 128, 96), (1, 96, 40, 64, 3)])
def test_torgb_fast_paths_match_generic(shape, math, dev):
    """ToRGB (1x1 + bias + clamp + upsample2d(skip)): split-K (small layers) and the LDS-weights kernel vs the generic kernel."""
    from nerffaceediting_amd import _lib, dense_ops as D
    N, H, W, cin, cout = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(N, H, W, cin, generator=g).to(dev)
    styles = (torch.randn(N, cin, generator=g) * 0.05).to(dev)
    weight = torch.randn(cout, cin, 1, 1, generator=g).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    skip = torch.randn(N, H // 2, W // 2, cout, generator=g).to(dev)
    packed, _ = D.conv_pack(weight)
    kw = dict(bias=bias, lrelu=False, act_gain=1.0, clamp=256.0, skip=skip, math=math)
    fast = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_1X1, **kw)
    D.FAST_PATH = False
    try:
        slow = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_1X1, **kw)
    finally:
        D.FAST_PATH = True
    assert float((fast - slow).abs().max()) <= 1e-5 * float(slow.abs().max())


def test_fp16_huge_styles_are_prenormalised(dev):
    """`conv_math='fp16'` gives a demodulated layer the reference's pre-normalisation (modulated_conv2d,
    networks_stylegan2.py:53-66): weights / max|w[o]| at pack time (the operand-range half of the reference's factor; its 1 / sqrt(I k k)
    guards an fp16 accumulator this library does not have), styles / max|s| per sample in the demodulation pass, the coefficient formed
    from the normalised values so that both cancel.  Round 5 rounded `activation x style` straight to fp16 and
    SATURATED at +-65504: a style of 1e6 gave finite but WRONG values.  Now a layer whose styles are scaled by 1, 1e3 and 1e6 - and a
    weight tensor scaled by 1e4, which would overflow fp16 on its own - returns the fp32-grade (split-bf16, raw styles) result within the
    fp16 operand bound, the same bound in every case: nothing saturates.  The pre-normalised pieces are checked against their
    definitions too."""
    from nerffaceediting_amd import _lib, dense_ops as D
    g = torch.Generator(device="cpu").manual_seed(77)
    N, H, cin, cout = 2, 32, 64, 64
    x = torch.randn(N, H, H, cin, generator=g).to(dev)
    w0 = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    bias = torch.zeros(cout, device=dev)
    for mode in (_lib.NFE_CONV_3X3, _lib.NFE_CONV_3X3_UP2):
        for wscale, sscale in ((1.0, 1.0), (1.0, 1e3), (1.0, 1e6), (1e4, 1.0), (1e4, 1e6)):
            w = w0 * wscale
            packed, wsq = D.conv_pack(w, math="fp16", prenormalize=True)
            st = ((torch.randn(N, cin, generator=g) * 0.5 + 1.0) * sscale).to(dev)
            dc, sn = D.conv_demod(st, wsq, prenormalize=True)
            # definitions (networks_stylegan2.py:55-56, 64-65)
            alpha = 1.0 / w.abs().amax(dim=(1, 2, 3), keepdim=True)
            assert float((sn - st / st.abs().amax(dim=1, keepdim=True)).abs().max()) <= 1e-6
            assert float((wsq - (w * alpha).square().sum(dim=(2, 3))).abs().max()) <= 1e-6 * float(wsq.abs().max())
            want_dc = ((sn[:, None, :] ** 2 * wsq[None]).sum(-1) + 1e-8).rsqrt()
            assert float(((dc - want_dc) / want_dc).abs().max()) <= 1e-5
            assert float(sn.abs().max()) == 1.0
            y = D.modulated_conv(x, sn, packed, cout, mode, bias, dcoef=dc, math="fp16")
            rp, rwsq = D.conv_pack(w)
            ref = D.modulated_conv(x, st, rp, cout, mode, bias, dcoef=D.conv_demod(st, rwsq), math="bf16x3")
            assert bool(torch.isfinite(y).all()), (wscale, sscale)
            err = float((y - ref).abs().max()) / float(ref.abs().max())
            assert err <= 4e-3, (mode, wscale, sscale, err)


_TORGB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nerffaceediting_amd import _lib, dense_ops as D
dev = torch.device("cuda:0")
out = {}
for math in ("bf16x3", "bf16", "fp16"):
    for k, (N, H, W, cin, cout, planes, with_skip) in enumerate(((2, 64, 64, 128, 96, False, True), (3, 32, 96, 256, 96, False, True), (2, 128, 128, 128, 96, True, True),
                                                                 (1, 64, 32, 64, 96, False, False), (2, 32, 32, 32, 32, False, True))):
        g = torch.Generator(device="cpu").manual_seed(40 + k)
        x = torch.randn(N, H, W, cin, generator=g).to(dev)
        styles = (torch.randn(N, cin, generator=g) * 0.05).to(dev)
        weight = torch.randn(cout, cin, 1, 1, generator=g).to(dev)
        bias = torch.randn(cout, generator=g).to(dev)
        skip = torch.randn(N, H // 2, W // 2, cout, generator=g).to(dev) if with_skip else None
        packed, _ = D.conv_pack(weight, math=math) if math == "fp16" else D.conv_pack(weight)
        y = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_1X1, bias=bias, lrelu=False, act_gain=1.0, clamp=256.0, skip=skip, math=math,
                             **({"out_planes": True} if planes else {}))
        out[f"{math}_{k}"] = y.cpu().numpy()
np.savez(sys.argv[2], **out)
"""


def test_coalesced_torgb_is_bit_identical_to_the_per_lane_form(dev, tmp_path):
    """Round 5: torgb_coalesced_kernel (every global access = four adjacent lanes on 64 contiguous bytes, re-ordering in LDS; rows of
    32 k pixels, 32 k output channels) against torgb_kernel (every lane fetches its own operand bytes): the SAME BITS - MFMA order,
    epilogue and skip-tap order are unchanged.  NFE_TORGB_COALESCED is read once per process: one child process per form.  Cases:
    96-channel image path with and without skip, the tri-plane output layout, non-square rows, one M-block, all three operand formats."""
    import os
    import subprocess
    import sys
    import inspect
    from nerffaceediting_amd import dense_ops as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "torgb.py"
    text = _TORGB_SCRIPT
    if "out_planes" not in inspect.signature(D.modulated_conv).parameters:
        text = text.replace('**({"out_planes": True} if planes else {})', "")
    if "math" not in inspect.signature(D.conv_pack).parameters:
        text = text.replace('D.conv_pack(weight, math=math) if math == "fp16" else D.conv_pack(weight)', 'D.conv_pack(weight)')
    script.write_text(text)
    res = {}
    for tag, env in (("per_lane", {"NFE_TORGB_COALESCED": "0"}), ("coalesced", {"NFE_TORGB_COALESCED": "1"})):
        f = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, str(script), root, str(f)], env=dict(os.environ, **env), timeout=600, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = np.load(f)
    keys = sorted(res["per_lane"].files)
    assert keys == sorted(res["coalesced"].files) and len(keys) >= 10
    for k in keys:
        a, b = res["per_lane"][k], res["coalesced"][k]
        assert a.shape == b.shape and np.isfinite(a).all() and np.array_equal(a, b), (k, float(np.abs(a - b).max()))


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
@pytest.mark.parametrize("shape,C,want_x", [((2, 64, 64, 64, 128), 3, True), ((1, 96, 128, 32, 256), 3, False), ((3, 32, 32, 128, 64), 4, True),
                                            ((2, 40, 72, 48, 128), 1, False), ((2, 8, 8, 64, 128), 3, True), ((3, 16, 16, 32, 64), 3, False)])
def test_fused_torgb_matches_separate_layers(shape, C, want_x, math, dev):
    """nfe_conv_args.rgb_*: the block's ToRGB (1x1 modulated conv without demodulation, bias, clamp) and the skip path
    img = upsample2d(img) + y (networks_stylegan2.py:450-457) evaluated in conv1's epilogue == conv1 followed by the ToRGB layer;
    with want_x False the fp32 activation is not produced at all.  Includes ragged tiles (40 x 72) and 2 / 4 M-block groups."""
    from nerffaceediting_amd import _lib, dense_ops as D
    N, H, W, cin, cout = shape
    g = torch.Generator(device="cpu").manual_seed(23)
    x = torch.randn(N, H, W, cin, generator=g).to(dev)
    styles = (torch.randn(N, cin, generator=g) * 0.5 + 1.0).to(dev)
    weight = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    noise = torch.randn(H, W, generator=g).to(dev)
    rw = torch.randn(C, cout, 1, 1, generator=g).to(dev)
    rs = (torch.randn(N, cout, generator=g) * 0.05).to(dev)
    rb = torch.randn(C, generator=g).to(dev)
    skip = torch.randn(N, H // 2, W // 2, C, generator=g).to(dev)
    packed, wsq = D.conv_pack(weight)
    dcoef = D.conv_demod(styles, wsq)
    kw = dict(bias=bias, dcoef=dcoef, noise=noise, noise_strength=0.3, lrelu=True, act_gain=2 ** 0.5, clamp=256.0, math=math)
    if W < 32 and math == "bf16x3":            # images narrower than a tile take the LDS-DMA path in plain bf16 only (conv3_eligible)
        assert not D.fuses_rgb(_lib.NFE_CONV_3X3, math, N, H, W, cin, cout, C)
        pytest.skip("split-bf16 keeps images narrower than 32 on the generic kernel")
    assert D.fuses_rgb(_lib.NFE_CONV_3X3, math, N, H, W, cin, cout, C)
    out, rgb = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3, rgb=(rw, rs, rb, skip, 256.0), want_out=want_x, **kw)
    ref_x = D.modulated_conv(x, styles, packed, cout, _lib.NFE_CONV_3X3, **kw)
    assert (out is not None) == want_x
    if want_x:
        assert torch.equal(out, ref_x)
    xr = ref_x.double() * rs.double()[:, None, None, :]                                # fp64 restatement of the ToRGB layer on ref_x
    y = torch.einsum("nhwc,oc->nhwo", xr, rw.double().reshape(C, cout)) + rb.double()
    y = y.clamp(-256, 256)
    up = torch.nn.functional.conv_transpose2d(skip.double().permute(0, 3, 1, 2).reshape(N * C, 1, H // 2, W // 2),
                                              (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])).double() / 16)[None, None].to(dev),
                                              stride=2, padding=1).reshape(N, C, H, W).permute(0, 2, 3, 1)
    ref = y + up
    assert rgb.shape == (N, H, W, C)
    assert float((rgb.double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)
    assert D.fuses_rgb(_lib.NFE_CONV_3X3, math, N, 8, 8, cin, cout, C) == (math == "bf16") and not D.fuses_rgb(_lib.NFE_CONV_3X3, math, N, H, W, cin, cout, 96)
    assert not D.fuses_rgb(_lib.NFE_CONV_3X3, math, N, 2, 2, cin, cout, C)


@pytest.mark.parametrize("tag", ["SuperresolutionHybrid8X.64", "SuperresolutionHybrid4X.64", "SuperresolutionHybrid4X.128",
                                 "SuperresolutionHybrid2X.96", "SuperresolutionHybridDeepfp32.64"])
def test_superresolution_variants(tag, dev):
    """The other SR heads (superresolution.py:29-155) against outputs of the reference classes; same state_dict names."""
    import zlib
    from nerffaceediting_amd.training import superresolution as sr
    from oracle.dense_params import params_by_name
    z = load("dense_sr_variants")
    name, in_res = tag.split(".")
    in_res = int(in_res)
    res = {"SuperresolutionHybrid8X": 512, "SuperresolutionHybrid4X": 256, "SuperresolutionHybrid2X": 128, "SuperresolutionHybridDeepfp32": 256}[name]
    kw = {} if name.endswith("Deepfp32") else dict(sr_antialias=(tag != "SuperresolutionHybrid4X.128"))
    net = getattr(sr, name)(channels=32, img_resolution=res, sr_num_fp16_res=4, **kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert sorted(shapes) == [str(k) for k in z[tag + ".keys"]]
    net = load_module(net, params_by_name(int(z["seed"]), shapes), dev)
    rng = np.random.RandomState(zlib.crc32(tag.encode()) & 0x7FFFFFFF)
    x = t(rng.randn(1, 32, in_res, in_res), dev); ws = t(rng.randn(1, 14, 512), dev)
    out = net(x[:, :3].contiguous(), x, ws, noise_mode="const")
    assert out.shape == (1, 3, res, res)
    e = maxerr(out[:, :, 1::4, 3::4], z[tag + ".out_s4"])
    print(tag, e)
    assert e <= 3e-4 and abs(float(out.double().mean()) - float(z[tag + ".out_mean"])) <= 1e-5
