"""Seeded parameter generators shared by the golden-vector script and the tests (TEST INFRASTRUCTURE).
Names follow the reference's state_dict; values are N(0,1) weights as the reference
initialises them plus NON-zero biases / noise strengths so every term of every layer is exercised."""
import numpy as np
import torch


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def mapping_params(seed, z_dim, c_dim, w_dim, num_layers=2, lr_mul=0.01):
    r = np.random.RandomState(seed)
    p = {"embed.weight": _t(r.randn(w_dim, c_dim)), "embed.bias": _t(r.randn(w_dim) * 0.1)}
    fin = z_dim + w_dim
    for i in range(num_layers):
        p[f"fc{i}.weight"] = _t(r.randn(w_dim, fin) / lr_mul)           # FullyConnectedLayer init: randn / lr_multiplier
        p[f"fc{i}.bias"] = _t(r.randn(w_dim) * 0.1 / lr_mul)
        fin = w_dim
    p["w_avg"] = _t(r.randn(w_dim) * 0.5)
    return p


def layer_params(seed, cin, cout, w_dim, res, k=3, torgb=False, prefix=""):
    r = np.random.RandomState(seed)
    p = {prefix + "weight": _t(r.randn(cout, cin, k, k)), prefix + "bias": _t(r.randn(cout) * 0.2),
         prefix + "affine.weight": _t(r.randn(cin, w_dim)), prefix + "affine.bias": _t(1.0 + r.randn(cin) * 0.1)}
    if not torgb:
        p[prefix + "noise_const"] = _t(r.randn(res, res))
        p[prefix + "noise_strength"] = torch.tensor(float(r.randn() * 0.3 + 0.2), dtype=torch.float32)
    return p


def block_params(seed, cin, cout, w_dim, res, img_channels, prefix=""):
    p = {}
    if cin == 0:
        p[prefix + "const"] = _t(np.random.RandomState(seed).randn(cout, res, res))
    else:
        p.update(layer_params(seed + 1, cin, cout, w_dim, res, prefix=prefix + "conv0."))
    p.update(layer_params(seed + 2, cout, cout, w_dim, res, prefix=prefix + "conv1."))
    p.update(layer_params(seed + 3, cout, img_channels, w_dim, res, k=1, torgb=True, prefix=prefix + "torgb."))
    return p


def synthesis_params(seed, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512):
    """SynthesisNetwork state (networks_stylegan2.py:469-501): blocks b4..b{img_resolution}."""
    p = {}
    res = 4
    while res <= img_resolution:
        cout = min(channel_base // res, channel_max)
        cin = min(channel_base // (res // 2), channel_max) if res > 4 else 0
        p.update(block_params(seed + 10 * res, cin, cout, w_dim, res, img_channels, prefix=f"b{res}."))
        res *= 2
    return p


def sr_params(seed):
    """SuperresolutionHybrid8XDC state (superresolution.py:264-277)."""
    p = block_params(seed, 32, 256, 512, 256, 3, prefix="block0.")
    p.update(block_params(seed + 100, 256, 128, 512, 512, 3, prefix="block1."))
    return p


def generator_params(seed, channel_base=4096, channel_max=32, z_dim=512, c_dim=25, w_dim=512):
    """Whole TriPlaneGenerator state (SURVEY.md App. B names) for a REDUCED backbone width (channel_base /
    channel_max) — the SR head and the decoder are always full size."""
    p = {}
    for k, v in mapping_params(seed, z_dim, c_dim, w_dim).items():
        p["backbone.mapping." + k] = v
    for k, v in synthesis_params(seed + 1, w_dim, 256, 96, channel_base, channel_max).items():
        p["backbone.synthesis." + k] = v
    for k, v in sr_params(seed + 2).items():
        p["superresolution." + k] = v
    r = np.random.RandomState(seed + 3)
    for net, out in (("geo_net", 16), ("app_net", 32)):
        p[f"decoder.{net}.0.weight"] = _t(r.randn(64, 32)); p[f"decoder.{net}.0.bias"] = _t(r.randn(64) * 0.2)
        p[f"decoder.{net}.2.weight"] = _t(r.randn(out, 64)); p[f"decoder.{net}.2.bias"] = _t(r.randn(out) * 0.2)
    # tone the backbone's output scale down so densities/colours are not saturated with random weights
    for k in list(p):
        if k.startswith("backbone.synthesis.") and k.endswith("torgb.weight"):
            p[k] = p[k] * 0.3
    return p


def params_by_name(seed, shapes):
    """Seeded parameters for any module of the dense path, keyed by state_dict name (order-independent: each tensor's
    generator is seeded with crc32(name) ^ seed).  shapes: {name: shape}; resample filters are left alone."""
    import zlib
    out = {}
    for k, shp in shapes.items():
        if k.endswith("resample_filter"):
            continue
        rng = np.random.RandomState((zlib.crc32(k.encode()) ^ int(seed)) & 0x7FFFFFFF)
        shp = tuple(shp)
        if k.endswith("affine.bias"):
            v = 1.0 + 0.2 * rng.randn(*shp)
        elif k.endswith("noise_strength"):
            v = 0.1 * rng.randn(*shp) if shp else np.float64(0.1 * rng.randn())
        elif k.endswith(".bias"):
            v = 0.1 * rng.randn(*shp)
        else:
            v = rng.randn(*shp)
        out[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shp).copy())
    return out
