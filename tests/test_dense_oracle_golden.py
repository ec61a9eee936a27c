"""CPU: the torch-CPU dense oracle against golden vectors captured from the reference's own modules
(oracle/gen_golden_dense.py).  Parameters are regenerated from the fixture's seeds."""
import numpy as np
import pytest
import torch

from oracle import dense_oracle as dor
from oracle.dense_params import layer_params, mapping_params, sr_params, synthesis_params
from tests._golden import load


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def err(a, b):
    return float((a.double() - t(b).double()).abs().max())


def test_mapping():
    z = load("dense_mapping")
    p = mapping_params(int(z["seed"]), int(z["z_dim"]), int(z["c_dim"]), int(z["w_dim"]))
    for tag, (psi, cut) in dict(a=(1.0, None), b=(0.7, None), c=(0.5, 4)).items():
        ws = dor.mapping(p, t(z["z"]), t(z["c"]), int(z["num_ws"]), 2, psi, cut)
        assert err(ws, z["ws." + tag]) <= 1e-5


LAYERS = ["conv_up1", "conv_up1_g", "conv_up2", "conv_up2_odd", "torgb96", "torgb3"]


def layer_case(z, tag):
    seed, cin, cout, res, up, clamp, gain = z[tag + ".cfg"]
    cfg = dict(seed=int(seed), cin=int(cin), cout=int(cout), res=int(res), up=int(up), clamp=None if clamp < 0 else float(clamp),
               gain=float(gain), torgb=tag.startswith("torgb"))
    p = layer_params(cfg["seed"], cfg["cin"], cfg["cout"], int(z["w_dim"]), cfg["res"], k=1 if cfg["torgb"] else 3, torgb=cfg["torgb"])
    return cfg, p


@pytest.mark.parametrize("tag", LAYERS)
def test_layers(tag):
    z = load("dense_layers")
    cfg, p = layer_case(z, tag)
    x, w = t(z[tag + ".x"]), t(z[tag + ".w"])
    if cfg["torgb"]:
        y = dor.torgb_layer(p, x, w, conv_clamp=cfg["clamp"])
    else:
        y = dor.synthesis_layer(p, x, w, up=cfg["up"], conv_clamp=cfg["clamp"], gain=cfg["gain"])
    assert err(y, z[tag + ".out"]) <= 2e-5


def test_upsample_and_resize():
    z = load("dense_layers")
    assert err(dor.upsample2d(t(z["upsample2d.x"])), z["upsample2d.out"]) <= 1e-6
    for tag in ("down_aa", "up_aa", "down_noaa", "odd_aa"):
        out = z[f"resize.{tag}.out"]
        y = dor.resize_bilinear(t(z[f"resize.{tag}.x"]), out.shape[2], out.shape[3], bool(int(z[f"resize.{tag}.aa"])))
        assert err(y, out) <= 2e-6, tag


def test_reduced_synthesis_network():
    z = load("dense_synthesis")
    p = synthesis_params(int(z["seed"]), int(z["w_dim"]), int(z["res"]), 96, int(z["channel_base"]), int(z["channel_max"]))
    res = [4, 8, 16, 32]
    assert err(dor.synthesis_network(p, t(z["ws"]), res), z["out"]) <= 5e-5


def test_full_width_synthesis_network():
    """The oracle at the FFHQ backbone width (512 channels, 256 px) against the sub-sampled reference output."""
    z = load("dense_synthesis_full")
    p = synthesis_params(int(z["seed"]), int(z["w_dim"]), int(z["res"]), 96, int(z["channel_base"]), int(z["channel_max"]))
    out = dor.synthesis_network(p, t(z["ws"]), [4, 8, 16, 32, 64, 128, 256])
    assert err(out[:, :, 3::8, 5::8], z["out_s8"]) <= 2e-4 * float(z["absmax"])
    assert err(out.mean(dim=(2, 3)), z["ch_mean"]) <= 1e-4


def test_superresolution_r64():
    z = load("dense_sr")
    p = sr_params(int(z["seed"]))
    x = t(z["r64.x"])
    y = dor.superresolution_8xdc(p, x[:, :3].contiguous(), x, t(z["r64.ws"]))
    assert err(y[:, :, ::4, ::4], z["r64.out_s4"]) <= 2e-4
    assert abs(float(y.mean()) - float(z["r64.out_mean"])) <= 1e-5
