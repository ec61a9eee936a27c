"""Tensor-level wrappers over include/nfe_dense.h (mapping network, modulated convs, resize, layouts).
Activations are NHWC fp32 between layers."""
import ctypes

import torch

from . import _lib
from .ops import _dev, _ptr, _stream

MATH = {None: _lib.NFE_CONV_BF16X3, "bf16x3": _lib.NFE_CONV_BF16X3, "bf16": _lib.NFE_CONV_BF16, "fp16": _lib.NFE_CONV_F16,
        _lib.NFE_CONV_BF16X3: _lib.NFE_CONV_BF16X3, _lib.NFE_CONV_BF16: _lib.NFE_CONV_BF16, _lib.NFE_CONV_F16: _lib.NFE_CONV_F16}


def _call(dev, rc_fn, what):
    with torch.cuda.device(dev):
        _lib.check(rc_fn(), what)


def nchw_to_nhwc(x):
    lib = _lib.load()
    x = _dev(x, "x", (None, None, None, None))
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, C, device=x.device)
    _call(x.device, lambda: lib.nfe_nchw_to_nhwc(_ptr(x), N, C, H, W, _ptr(out), _stream()), "nfe_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x):
    if x.requires_grad and torch.is_grad_enabled():      # part of an autograd graph (plane editing): a view keeps it differentiable
        return x.permute(0, 3, 1, 2).contiguous()
    lib = _lib.load()
    x = _dev(x, "x", (None, None, None, None))
    N, H, W, C = x.shape
    out = torch.empty(N, C, H, W, device=x.device)
    _call(x.device, lambda: lib.nfe_nhwc_to_nchw(_ptr(x), N, C, H, W, _ptr(out), _stream()), "nfe_nhwc_to_nchw")
    return out


def nhwc_to_planes(x):
    """[N,H,W,96] -> tri-plane gather layout [N,3,H,W,32]."""
    lib = _lib.load()
    x = _dev(x, "planes", (None, None, None, 96))
    N, H, W, _ = x.shape
    out = torch.empty(N, 3, H, W, 32, device=x.device)
    _call(x.device, lambda: lib.nfe_nhwc_to_planes(_ptr(x), N, H, W, _ptr(out), _stream()), "nfe_nhwc_to_planes")
    return out


def plane_stats_nhwc(x):
    """compute_mean_var on NHWC [N,H,W,C] -> mean, std [N,C,1,1]."""
    lib = _lib.load()
    x = _dev(x, "planes", (None, None, None, None))
    N, H, W, C = x.shape
    mean = torch.empty(N, C, 1, 1, device=x.device)
    std = torch.empty_like(mean)
    scratch = torch.empty(N * C * 2, dtype=torch.float64, device=x.device)
    _call(x.device, lambda: lib.nfe_plane_stats_nhwc(_ptr(x), N, H * W, C, _ptr(mean), _ptr(std), _ptr(scratch), _stream()),
          "nfe_plane_stats_nhwc")
    return mean, std


def fully_connected(x, weight, bias, weight_gain, bias_gain=1.0, lrelu=False, out=None, out_offset=0):
    """FullyConnectedLayer.forward; `out` (a [N,S] buffer) + out_offset write into a wider row (concat)."""
    lib = _lib.load()
    x = _dev(x, "x", (None, None))
    N, fin = x.shape
    weight = _dev(weight, "weight", (None, fin))
    fout = weight.shape[0]
    if bias is not None:
        bias = _dev(bias, "bias", (fout,))
    if out is None:
        out = torch.empty(N, fout, device=x.device)
        view, stride = out, fout
    else:
        stride = out.shape[1]
        view = out[:, out_offset:]
    _call(x.device, lambda: lib.nfe_fully_connected(_ptr(x), _ptr(weight), _ptr(bias), N, fin, fout, float(weight_gain),
                                                    float(bias_gain), int(bool(lrelu)),
                                                    ctypes.c_void_p(view.data_ptr()), stride, _stream()), "nfe_fully_connected")
    return out


def fully_connected_grouped(groups):
    """nfe_fully_connected_grouped: groups = [(x [N,in] (rows may be strided: a column block of ws), weight, bias, weight_gain,
    bias_gain)] -> list of [N,out] tensors, one launch for all of them."""
    lib = _lib.load()
    outs, keep = [], []
    arr = (_lib.FcGroup * len(groups))()
    N, dev = groups[0][0].shape[0], groups[0][0].device
    for i, (x, w, b, wg, bg) in enumerate(groups):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] == N and x.stride(1) == 1
        w = _dev(w, "weight", (None, x.shape[1]))
        y = torch.empty(N, w.shape[0], device=dev)
        g = arr[i]
        g.x, g.x_stride, g.w, g.b, g.y = x.data_ptr(), x.stride(0) if N > 1 else x.shape[1], w.data_ptr(), (b.data_ptr() if b is not None else None), y.data_ptr()
        g.in_features, g.out_features, g.weight_gain, g.bias_gain = x.shape[1], w.shape[0], float(wg), float(bg)
        outs.append(y); keep += [x, w, b]
    _call(dev, lambda: lib.nfe_fully_connected_grouped(arr, len(groups), N, _stream()), "nfe_fully_connected_grouped")
    return outs


def conv_demod_grouped(pairs, prenormalize=False):
    """nfe_conv_demod_grouped: pairs = [(styles [N,cin], wsq [cout,cin])] -> list of dcoef [N,cout], one launch.
    prenormalize (fp16 operand mode) -> (list of dcoef, list of pre-normalised styles): see conv_demod."""
    lib = _lib.load()
    arr = (_lib.DemodGroup * len(pairs))()
    N, dev = pairs[0][0].shape[0], pairs[0][0].device
    outs, norms = [], []
    for i, (s, wsq) in enumerate(pairs):
        d = torch.empty(N, wsq.shape[0], device=dev)
        g = arr[i]
        g.styles, g.wsq, g.dcoef, g.cin, g.cout = s.data_ptr(), wsq.data_ptr(), d.data_ptr(), s.shape[1], wsq.shape[0]
        if prenormalize:
            norms.append(torch.empty_like(s))
            g.styles_norm = norms[-1].data_ptr()
        outs.append(d)
    _call(dev, lambda: lib.nfe_conv_demod_grouped(arr, len(pairs), N, _stream()), "nfe_conv_demod_grouped")
    return (outs, norms) if prenormalize else outs


def normalize_2nd_moment(x, out=None, out_offset=0):
    lib = _lib.load()
    x = _dev(x, "x", (None, None))
    N, f = x.shape
    if out is None:
        out = torch.empty_like(x)
        view, stride = out, f
    else:
        stride = out.shape[1]
        view = out[:, out_offset:]
    _call(x.device, lambda: lib.nfe_normalize_2nd_moment(_ptr(x), N, f, ctypes.c_void_p(view.data_ptr()), stride, _stream()),
          "nfe_normalize_2nd_moment")
    return out


def broadcast_truncate(w, w_avg, num_ws, psi=1.0, cutoff=None):
    lib = _lib.load()
    w = _dev(w, "w", (None, None))
    N, D = w.shape
    if w_avg is not None:
        w_avg = _dev(w_avg, "w_avg", (D,))
    cutoff = num_ws if cutoff is None else int(cutoff)
    ws = torch.empty(N, num_ws, D, device=w.device)
    _call(w.device, lambda: lib.nfe_broadcast_truncate(_ptr(w), _ptr(w_avg), N, D, num_ws, float(psi), cutoff, _ptr(ws), _stream()),
          "nfe_broadcast_truncate")
    return ws


def conv_pack(weight, math=None, prenormalize=False):
    """[Cout,Cin,k,k] -> (packed MFMA fragment image, wsq [Cout,Cin]).  math='fp16' packs fp16 operand words (for layers run with
    math='fp16' only); every other mode shares the bf16 hi + lo image.  prenormalize (fp16, demodulated layers): the reference's
    weight pre-normalisation (networks_stylegan2.py:55); pair it with conv_demod(.., prenormalize=True)."""
    lib = _lib.load()
    weight = _dev(weight, "weight", (None, None, None, None))
    cout, cin, k, _ = weight.shape
    words = lib.nfe_conv_packed_words(cout, cin, k)
    if words == 0:
        raise RuntimeError(f"conv_pack: unsupported weight shape {list(weight.shape)}")
    f16 = MATH[math] == _lib.NFE_CONV_F16
    assert f16 or not prenormalize, "prenormalize belongs to the fp16 operand mode"
    packed = torch.empty(words + (cout if f16 else 0), device=weight.device)
    wsq = torch.empty(cout, cin, device=weight.device)
    if f16:
        _call(weight.device, lambda: lib.nfe_conv_pack_f16(_ptr(weight), cout, cin, k, int(bool(prenormalize)), _ptr(packed), _ptr(wsq), _stream()),
              "nfe_conv_pack_f16")
    else:
        _call(weight.device, lambda: lib.nfe_conv_pack(_ptr(weight), cout, cin, k, _ptr(packed), _ptr(wsq), _stream()), "nfe_conv_pack")
    return packed, wsq


def conv_demod(styles, wsq, prenormalize=False):
    """dcoef [N,Cout].  prenormalize (fp16 operand mode): -> (dcoef, styles / max|styles| per sample), the reference's style
    pre-normalisation (networks_stylegan2.py:56); wsq must come from conv_pack(.., 'fp16', prenormalize=True)."""
    lib = _lib.load()
    styles = _dev(styles, "styles", (None, None))
    N, cin = styles.shape
    wsq = _dev(wsq, "wsq", (None, cin))
    cout = wsq.shape[0]
    d = torch.empty(N, cout, device=styles.device)
    sn = torch.empty_like(styles) if prenormalize else None
    _call(styles.device, lambda: lib.nfe_conv_demod(_ptr(styles), _ptr(wsq), N, cin, cout, _ptr(d), _ptr(sn), _stream()), "nfe_conv_demod")
    return (d, sn) if prenormalize else d


FAST_PATH = True      # tests flip this to compare the LDS-DMA 3x3 path with the generic kernel


class SplitImage:
    """bf16 hi(+lo) image of a modulated NHWC activation, produced by one layer for the 3x3 layer that consumes it."""

    def __init__(self, data, shape, math):
        self.data, self.shape, self.math = data, tuple(shape), math


def can_chain(mode, math, n, h, w, cin, cout):
    """True if a layer of these sizes runs the fast path and can therefore take a SplitImage input."""
    if not FAST_PATH:
        return False
    return bool(_lib.load().nfe_conv_accepts_split(int(mode), h, w, cin, cout))


def fuses_rgb(mode, math, n, h, w, cin, cout, rgb_channels):
    """True if a 3x3 layer of these sizes can evaluate the block's ToRGB in its epilogue (nfe_conv_fuses_rgb)."""
    if not FAST_PATH:
        return False
    return bool(_lib.load().nfe_conv_fuses_rgb(int(mode), MATH[math], n, h, w, cin, cout, rgb_channels))


def splits_in_epilogue(mode, n, h, w, cin, cout):
    """True if a plain 3x3 layer of these sizes writes the consuming layer's SplitImage from its own epilogue
    (nfe_conv_splits_in_epilogue): the fp32 output may then be skipped (want_out=False) when nobody else reads it."""
    if not FAST_PATH:
        return False
    return bool(_lib.load().nfe_conv_splits_in_epilogue(int(mode), n, h, w, cin, cout))


def describe(mode, math, n, h, w, cin, cout, rgb_channels=0):
    """nfe_conv_describe: which kernels a layer of these sizes runs at batch n (diagnostic text)."""
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.load().nfe_conv_describe(int(mode), MATH[math], n, h, w, cin, cout, int(rgb_channels), buf, 256), "nfe_conv_describe")
    return buf.value.decode()


def modulated_conv(x, styles, packed, cout, mode, bias, dcoef=None, noise=None, noise_strength=0.0, lrelu=True,
                   act_gain=1.0, clamp=None, skip=None, out_planes=False, math=None, next_styles=None, want_out=True, rgb=None):
    """nfe_modulated_conv.  x [N,H,W,Cin] NHWC (or a SplitImage of the modulated input) -> [N,Ho,Wo,Cout]
    (or [N,3,Ho,Wo,32] if out_planes).  With next_styles [N,Cout] also returns the SplitImage for the consuming
    3x3 layer: (out, split); out is None if want_out is False (up-sampling layers, and plain ones where splits_in_epilogue()).
    rgb = (weight [C,Cout], styles [N,Cout], bias [C], skip [N,H/2,W/2,C] or None, clamp): the block's ToRGB evaluated in
    this layer's epilogue (only where fuses_rgb() is true) -> returns (out or None, rgb_image [N,H,W,C]) (+ split with next_styles)."""
    lib = _lib.load()
    a = _lib.ConvArgs()
    a.struct_size = ctypes.sizeof(_lib.ConvArgs)
    a.mode, a.math = int(mode), MATH[math]
    if isinstance(x, SplitImage):
        assert x.math == MATH[math], "SplitImage was produced for another math mode"
        N, H, W, cin = x.shape
        a.x_split = x.data.data_ptr()
        dev = x.data.device
    else:
        x = _dev(x, "x", (None, None, None, None))
        N, H, W, cin = x.shape
        a.x = x.data_ptr()
        dev = x.device
    styles = _dev(styles, "styles", (N, cin))
    a.styles, a.packed = styles.data_ptr(), _dev(packed, "packed").data_ptr()
    keep = [x, styles, packed]
    if dcoef is not None:
        dcoef = _dev(dcoef, "dcoef", (N, cout)); a.dcoef = dcoef.data_ptr()
    up = 2 if mode == _lib.NFE_CONV_3X3_UP2 else 1
    Ho, Wo = H * up, W * up
    if noise is not None:
        noise = _dev(noise, "noise")
        assert tuple(noise.shape[-2:]) == (Ho, Wo) and noise.numel() in (Ho * Wo, N * Ho * Wo), "noise must be [Ho,Wo] or [N,1,Ho,Wo]"
        a.noise, a.noise_strength = noise.data_ptr(), float(noise_strength)
        a.noise_n_stride = Ho * Wo if noise.numel() == N * Ho * Wo and N > 1 else 0
    bias = _dev(bias, "bias", (cout,)); a.bias = bias.data_ptr()
    a.n, a.h, a.w, a.cin, a.cout = N, H, W, cin, cout
    a.lrelu, a.act_gain = int(bool(lrelu)), float(act_gain)
    a.clamp = -1.0 if clamp is None else float(clamp)
    if skip is not None:
        skip = _dev(skip, "skip", (N, H // 2, W // 2, cout)); a.skip = skip.data_ptr()
    a.out_planes = int(bool(out_planes))
    out = None
    rgb_out = None
    if rgb is not None:
        rw, rs, rb, rskip, rclamp = rgb
        C = rw.shape[0]
        rw = _dev(rw.reshape(C, cout), "rgb_weight", (C, cout)); rs = _dev(rs, "rgb_styles", (N, cout)); rb = _dev(rb, "rgb_bias", (C,))
        rgb_out = torch.empty(N, Ho, Wo, C, device=dev)
        a.rgb_weight, a.rgb_styles, a.rgb_bias, a.rgb_out, a.rgb_channels = rw.data_ptr(), rs.data_ptr(), rb.data_ptr(), rgb_out.data_ptr(), C
        a.rgb_clamp = -1.0 if rclamp is None else float(rclamp)
        if rskip is not None:
            rskip = _dev(rskip, "rgb_skip", (N, H // 2, W // 2, C)); a.rgb_skip = rskip.data_ptr()
        keep += [rw, rs, rb, rskip, rgb_out]
    if want_out or (next_styles is None and rgb is None):
        out = torch.empty((N, 3, Ho, Wo, 32) if out_planes else (N, Ho, Wo, cout), device=dev)
        a.out = out.data_ptr()
    split = None
    if next_styles is not None:
        next_styles = _dev(next_styles, "next_styles", (N, cout))
        split = SplitImage(torch.empty(int(lib.nfe_conv_split_floats(a.math, N, Ho, Wo, cout)), device=dev), (N, Ho, Wo, cout), a.math)
        a.next_styles, a.next_split = next_styles.data_ptr(), split.data.data_ptr()
    scratch = None
    n_scratch = int(lib.nfe_conv_scratch_floats(a.mode, a.math, N, H, W, cin, cout)) if FAST_PATH else (N * (2 * H + 1) * (2 * W + 1) * cout if up == 2 else 0)
    if n_scratch:                    # up-conv intermediate, or the pre-split input image of the 3x3 fast path
        scratch = torch.empty(n_scratch, device=dev)
        a.scratch, a.scratch_floats = scratch.data_ptr(), n_scratch
    keep += [dcoef, noise, bias, skip, scratch, next_styles, split]
    _call(dev, lambda: lib.nfe_modulated_conv(ctypes.byref(a), _stream()), "nfe_modulated_conv")
    if rgb is not None:
        return (out, rgb_out) if next_styles is None else (out, rgb_out, split)
    return out if next_styles is None else (out, split)


def upfirdn2d(x, up=1, down=1, padding=(0, 0), gain=1.0):
    """upfirdn2d (torch_utils/ops/upfirdn2d.py:120) with the path's [1,3,3,1] filter on NHWC: x [N,H,W,C] -> [N,OH,OW,C]."""
    lib = _lib.load()
    x = _dev(x, "x", (None, None, None, None))
    N, H, W, C = x.shape
    p0, p1 = int(padding[0]), int(padding[1])
    oh, ow = (H * up + p0 + p1 - 4) // down + 1, (W * up + p0 + p1 - 4) // down + 1
    out = torch.empty(N, oh, ow, C, device=x.device)
    _call(x.device, lambda: lib.nfe_upfirdn2d(_ptr(x), N, H, W, C, int(up), int(down), p0, p1, float(gain), _ptr(out), _stream()), "nfe_upfirdn2d")
    return out


def upsample2d(x):
    """upfirdn2d.upsample2d(x, f, up=2) (upfirdn2d.py:315-350): the skip path of SynthesisBlock (networks_stylegan2.py:453)."""
    return upfirdn2d(x, up=2, padding=(2, 1), gain=4.0)


def resize_bilinear_backward(g, h, w, antialias=True):
    """Adjoint of resize_bilinear: g [N,OH,OW,C] (gradient of the resized image) -> [N,h,w,C] (gradient of its input)."""
    lib = _lib.load()
    g = _dev(g, "g", (None, None, None, None))
    N, OH, OW, C = g.shape
    out = torch.empty(N, int(h), int(w), C, device=g.device)
    _call(g.device, lambda: lib.nfe_resize_bilinear_backward(_ptr(g), N, int(h), int(w), C, OH, OW, int(bool(antialias)), _ptr(out), _stream()),
          "nfe_resize_bilinear_backward")
    return out


def upfirdn2d_polyphase(x, padding=(2, 2), gain=1.0):
    """upfirdn2d(x, up=1, down=1, padding, gain) [N,OH,OW,C] delivered as its four polyphase images stacked along the channels:
    [N, ceil(OH/2), ceil(OW/2), 4C], zeros beyond OH / OW (include/nfe_dense.h: nfe_upfirdn2d_polyphase)."""
    lib = _lib.load()
    x = _dev(x, "x", (None, None, None, None))
    N, H, W, C = x.shape
    OH, OW = H + padding[0] + padding[1] - 3, W + padding[0] + padding[1] - 3
    out = torch.empty(N, (OH + 1) // 2, (OW + 1) // 2, 4 * C, device=x.device)
    _call(x.device, lambda: lib.nfe_upfirdn2d_polyphase(_ptr(x), N, H, W, C, int(padding[0]), int(padding[1]), float(gain), _ptr(out), _stream()),
          "nfe_upfirdn2d_polyphase")
    return out


def bias_act_backward(out, grad=None, grad_rgb=None, rgb_w=None, rgb_s=None, scale=None, gain=1.0, clamp=None):
    """Backward of bias_act('lrelu', gain, clamp) (bias_act.py:93-125) at a layer's saved output `out` [N,H,W,C], in ONE pass, with the
    block's transposed ToRGB folded in (include/nfe_dense.h: nfe_bias_act_backward):
    ((grad or 0) + (grad_rgb @ rgb_w) * rgb_s) * gain * (0.2 if out < 0 else 1) * (|out| < clamp) * (scale or 1).
    grad [N,H,W,C] or None; grad_rgb [N,H,W,K] with rgb_w [K,C], rgb_s [N,C] or None; scale [N,C] or None."""
    lib = _lib.load()
    out = _dev(out, "out", (None, None, None, None))
    N, H, W, C = out.shape
    if grad is not None:
        grad = _dev(grad.contiguous(), "grad", (N, H, W, C))
    K = 0
    if grad_rgb is not None:
        K = int(grad_rgb.shape[-1])
        grad_rgb = _dev(grad_rgb.contiguous(), "grad_rgb", (N, H, W, K))
        rgb_w = _dev(rgb_w.contiguous(), "rgb_w", (K, C))
        rgb_s = _dev(rgb_s.contiguous(), "rgb_s", (N, C))
    if scale is not None:
        scale = _dev(scale.contiguous(), "scale", (N, C))
    dst = torch.empty_like(out)
    _call(out.device, lambda: lib.nfe_bias_act_backward(_ptr(out), _ptr(grad), _ptr(grad_rgb), _ptr(rgb_w), _ptr(rgb_s), K, _ptr(scale), float(gain),
                                                        float(clamp) if clamp is not None else 0.0, N, H * W, C, _ptr(dst), _stream()),
          "nfe_bias_act_backward")
    return dst


def resize_bilinear(x, oh, ow, antialias=True):
    """F.interpolate(mode='bilinear', align_corners=False, antialias=...) on NHWC."""
    lib = _lib.load()
    x = _dev(x, "x", (None, None, None, None))
    N, H, W, C = x.shape
    out = torch.empty(N, oh, ow, C, device=x.device)
    _call(x.device, lambda: lib.nfe_resize_bilinear(_ptr(x), N, H, W, C, int(oh), int(ow), int(bool(antialias)), _ptr(out), _stream()),
          "nfe_resize_bilinear")
    return out
