"""CPU oracle for the dense half of the path (mapping network, StyleGAN2 synthesis blocks,
SuperresolutionHybrid8XDC) — explicit-formula restatement of SURVEY.md App. A.6, with torch-CPU as the
array library (conv2d / conv_transpose2d are the same ATen arithmetic the reference runs on CPU).

TEST INFRASTRUCTURE ONLY: imported by tests/, smoke() and bench.py's cpu_baseline leg, never by the
product path.  Pinned by oracle/gen_golden_dense.py against the reference imported from /root/reference
(fixtures under tests/golden/dense_*.npz).

Every function takes/returns torch CPU float32 tensors in the reference's NCHW layout and cites the
reference file:line it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def fully_connected(x, weight, bias, lr_mul=1.0, act="linear"):
    """FullyConnectedLayer.forward, networks_stylegan2.py:114-127."""
    w = weight * (lr_mul / math.sqrt(weight.shape[1]))
    y = x @ w.t()
    if bias is not None:
        y = y + bias * lr_mul
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2) * SQRT2
    return y


def normalize_2nd_moment(x, eps=1e-8):
    """networks_stylegan2.py:24-26."""
    return x * (x.square().mean(dim=1, keepdim=True) + eps).rsqrt()


def mapping(p, z, c, num_ws, num_layers=2, truncation_psi=1.0, truncation_cutoff=None, lr_mul=0.01):
    """MappingNetwork.forward, networks_stylegan2.py:233-268.  p: dict of embed/fc{i} weights+biases, w_avg."""
    x = normalize_2nd_moment(z)
    y = normalize_2nd_moment(fully_connected(c, p["embed.weight"], p["embed.bias"]))
    x = torch.cat([x, y], dim=1)
    for i in range(num_layers):
        x = fully_connected(x, p[f"fc{i}.weight"], p[f"fc{i}.bias"], lr_mul=lr_mul, act="lrelu")
    x = x.unsqueeze(1).repeat(1, num_ws, 1)
    if truncation_psi != 1:
        if truncation_cutoff is None:
            x = p["w_avg"].lerp(x, truncation_psi)
        else:
            x[:, :truncation_cutoff] = p["w_avg"].lerp(x[:, :truncation_cutoff], truncation_psi)
    return x


def _fir():
    f = torch.tensor([1.0, 3.0, 3.0, 1.0])
    f = torch.outer(f, f)
    return f / f.sum()                                      # upfirdn2d.setup_filter, upfirdn2d.py:72-116


def _upfirdn_pad_filter(x, pad, gain):
    """upfirdn2d(x, f, padding=pad, gain=gain) for up=down=1 (upfirdn2d.py:169-207): pad, true convolution."""
    C = x.shape[1]
    f = (_fir() * gain).flip([0, 1])[None, None].repeat(C, 1, 1, 1)
    x = F.pad(x, [pad[0], pad[1], pad[2], pad[3]])
    return F.conv2d(x, f, groups=C)


def upsample2d(x):
    """upfirdn2d.upsample2d(x, f, up=2), upfirdn2d.py:315-350: zero-insert, pad (2,1), FIR*4."""
    N, C, H, W = x.shape
    u = torch.zeros(N, C, H * 2, W * 2)
    u[:, :, ::2, ::2] = x
    return _upfirdn_pad_filter(u, (2, 1, 2, 1), 4.0)


def modulated_conv(x, weight, styles, noise=None, up=1, demodulate=True):
    """modulated_conv2d fused path, networks_stylegan2.py:58-91 + conv2d_resample.py:48-143."""
    N = x.shape[0]
    Co, Ci, kh, kw = weight.shape
    w = weight[None] * styles.reshape(N, 1, Ci, 1, 1)
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(N, Co, 1, 1, 1)
    outs = []
    for n in range(N):
        if up == 1:
            y = F.conv2d(x[n:n + 1], w[n], padding=kh // 2)
        else:                                               # conv2d_resample.py:114-128
            y = F.conv_transpose2d(x[n:n + 1], w[n].transpose(0, 1), stride=2, padding=0)
            y = _upfirdn_pad_filter(y, (1, 1, 1, 1), 4.0)
        outs.append(y)
    y = torch.cat(outs, 0)
    if noise is not None:
        y = y + noise
    return y


def bias_act(x, b, act="linear", gain=None, clamp=None):
    """_bias_act_ref, bias_act.py:93-125."""
    x = x + b.reshape(1, -1, 1, 1)
    if act == "lrelu":
        x = F.leaky_relu(x, 0.2)
        gain = SQRT2 if gain is None else gain
    gain = 1.0 if gain is None else gain
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def synthesis_layer(p, x, w, up=1, noise_mode="const", conv_clamp=None, gain=1.0):
    """SynthesisLayer.forward, networks_stylegan2.py:311-330.  p: weight, bias, affine.weight, affine.bias,
    noise_const, noise_strength."""
    styles = fully_connected(w, p["affine.weight"], p["affine.bias"])
    noise = p["noise_const"] * p["noise_strength"] if noise_mode == "const" else None
    x = modulated_conv(x, p["weight"], styles, noise=noise, up=up)
    return bias_act(x, p["bias"], act="lrelu", gain=SQRT2 * gain, clamp=None if conv_clamp is None else conv_clamp * gain)


def torgb_layer(p, x, w, conv_clamp=None):
    """ToRGBLayer.forward, networks_stylegan2.py:353-357."""
    Ci = p["weight"].shape[1]
    styles = fully_connected(w, p["affine.weight"], p["affine.bias"]) * (1 / math.sqrt(Ci))
    x = modulated_conv(x, p["weight"], styles, demodulate=False)
    return bias_act(x, p["bias"], clamp=conv_clamp)


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def synthesis_block(p, x, img, ws, first=False, noise_mode="const", conv_clamp=None):
    """SynthesisBlock.forward (skip architecture), networks_stylegan2.py:417-461.  ws [N, num_conv+1, w_dim]."""
    it = iter(ws.unbind(dim=1))
    if first:
        x = p["const"][None].repeat(ws.shape[0], 1, 1, 1)
    else:
        x = synthesis_layer(_sub(p, "conv0."), x, next(it), up=2, noise_mode=noise_mode, conv_clamp=conv_clamp)
    x = synthesis_layer(_sub(p, "conv1."), x, next(it), noise_mode=noise_mode, conv_clamp=conv_clamp)
    if img is not None:
        img = upsample2d(img)
    y = torgb_layer(_sub(p, "torgb."), x, next(it), conv_clamp=conv_clamp)
    img = img + y if img is not None else y
    return x, img


def synthesis_network(p, ws, resolutions, noise_mode="const"):
    """SynthesisNetwork.forward, networks_stylegan2.py:503-518 (block k consumes ws[idx:idx+num_conv+1])."""
    x = img = None
    idx = 0
    for res in resolutions:
        first = res == resolutions[0]
        nconv = 1 if first else 2
        x, img = synthesis_block(_sub(p, f"b{res}."), x, img, ws[:, idx:idx + nconv + 1], first=first, noise_mode=noise_mode)
        idx += nconv
    return img


def _aa_weights(in_size, out_size, antialias):
    """ATen UpSampleKernel.cpp compute_indices_weights_aa (bilinear: interp_size 2) / plain bilinear."""
    scale = in_size / out_size
    W = np.zeros((out_size, in_size), dtype=np.float64)
    for i in range(out_size):
        if antialias:
            support = scale if scale >= 1.0 else 1.0
            inv = 1.0 / scale if scale >= 1.0 else 1.0
            center = scale * (i + 0.5)
            lo = max(int(center - support + 0.5), 0)
            n = min(int(center + support + 0.5), in_size) - lo
            w = np.array([max(0.0, 1.0 - abs((k + lo - center + 0.5) * inv)) for k in range(n)])
            W[i, lo:lo + n] = w / w.sum()
        else:
            src = max(scale * (i + 0.5) - 0.5, 0.0)
            lo = min(int(src), in_size - 1)
            lam = src - lo
            W[i, lo] += 1.0 - lam
            if lo + 1 < in_size:
                W[i, lo + 1] += lam
            else:
                W[i, lo] += lam
    return torch.from_numpy(W.astype(np.float32))


def resize_bilinear(x, oh, ow, antialias=True):
    """F.interpolate(x, (oh,ow), mode='bilinear', align_corners=False, antialias=...), superresolution.py:283-286."""
    Wy = _aa_weights(x.shape[2], oh, antialias)
    Wx = _aa_weights(x.shape[3], ow, antialias)
    return torch.einsum("oy,ncyx,px->ncop", Wy, x, Wx)


def superresolution_8xdc(p, rgb, x, ws, noise_mode="none", sr_antialias=True):
    """SuperresolutionHybrid8XDC.forward, superresolution.py:279-290 (conv_clamp 256 since sr_num_fp16_res>0)."""
    ws = ws[:, -1:, :].repeat(1, 3, 1)
    if x.shape[-1] != 128:
        x = resize_bilinear(x, 128, 128, sr_antialias)
        rgb = resize_bilinear(rgb, 128, 128, sr_antialias)
    x, rgb = synthesis_block(_sub(p, "block0."), x, rgb, ws, noise_mode=noise_mode, conv_clamp=256)
    x, rgb = synthesis_block(_sub(p, "block1."), x, rgb, ws, noise_mode=noise_mode, conv_clamp=256)
    return rgb
