"""Input gradient of the super-resolution head: d image / d (neural-rendered feature image), weights and ws constant.

The reference's `utils.decode` is differentiable end to end, the super-resolved `image` included (utils.py:165-199,
training/superresolution.py:279-290): an editing loss may mix a segmentation term with an image term.  Round 2 tied `image` to
a node that raised in backward.  Here the gradient is computed by the SAME hand-written MFMA kernels as the forward, called
through `nfe_modulated_conv` with re-packed weights:

  plain 3x3 layer  y = act((W * (s x)) d + noise + b):   dL/dx = s . [ W^T_flipped * (d . g_pre) ]            -> one NFE_CONV_3X3 call,
                                                         styles := d, dcoef := s, bias 0, no activation
  up-sampling layer (transposed conv stride 2, then the 4x4 FIR with pad (1,1), conv2d_resample.py:114-128):
                   g_T = FIR^T (d . g_pre)  = nfe_upfirdn2d(pad=(2,2), gain=4)            [2H+1]^2
                   dL/dx[y,x] = s . sum_{kh,kw} W[:, :, kh, kw]^T g_T[2y+kh, 2x+kw]  = a 2x2-tap convolution over the four
                   polyphase images of g_T stacked along the channels, embedded in one NFE_CONV_3X3 call (taps (1+dy, 1+dx))
  ToRGB 1x1        dL/dx = s . (g_y W)        (K = 3: a [N,H,W,3] x [3,C] product, torch glue)
  skip path        img = upsample2d(img_prev) + y:   dL/dimg_prev = nfe_upfirdn2d(g, down=2, pad=(1,2), gain=4)
  activation       g_pre = g_out * gain * (0.2 if out < 0 else 1) * (|out| < clamp), from the saved layer outputs

Elementwise masks, the polyphase re-arrangement and the K = 3 product are torch tensor expressions (glue); every convolution
and FIR runs in libnfe_render.so.  The forward of this path evaluates the head layer by layer and keeps the six activations
(~0.4 GB per view at 512^2): it is the editing path (a few views), not the throughput path, which is unchanged.
"""
import numpy as np
import torch

from . import _lib, dense_ops
from .training.networks_stylegan2 import _publish, batch_styles, block_layers

GRAD_MATH = "bf16x3"      # fp32-grade split-bf16 MFMA for every backward convolution


HEADS = ("SuperresolutionHybrid8XDC", "SuperresolutionHybrid8X", "SuperresolutionHybrid4X", "SuperresolutionHybrid2X",
         "SuperresolutionHybridDeepfp32")


def supported(sr, resolution):
    """True where SRImage can run: every two-block head of the reference (superresolution.py:29-290), fed at any neural rendering
    resolution (the bilinear / antialiased pre-resize to the head's input has its adjoint in nfe_resize_bilinear_backward)."""
    return type(sr).__name__ in HEADS and int(resolution) > 0


def _resizes(sr, r_in):
    """Whether the head resizes a feature image of this size before block0 (8XDC / 8X / 2X: whenever it differs; 4X / Deepfp32:
    only when it is smaller: superresolution.py:80, :145, :283)."""
    r = sr.input_resolution
    return (r_in < r) if getattr(sr, "resize_if_smaller_only", False) else (r_in != r)


def _act_grad(out, g, gain, clamp, rgb=None, scale=None):
    """d bias_act('lrelu', gain, clamp) / d pre-activation, applied to the incoming gradient (bias_act.py:93-125) - one HIP pass
    (nfe_bias_act_backward) that also adds the block's transposed ToRGB branch (rgb = (g_y [N,H,W,3], W^T [3,C], styles [N,C])) and
    applies a per-(view, channel) factor for the consumer; g may be None when only the ToRGB branch feeds the layer."""
    g_rgb, w_rgb, s_rgb = rgb if rgb is not None else (None, None, None)
    return dense_ops.bias_act_backward(out, grad=g, grad_rgb=g_rgb, rgb_w=w_rgb, rgb_s=s_rgb, scale=scale, gain=gain, clamp=clamp)


def _bwd_plain(layer):
    """Packed fragments of the backward-data kernel of a plain 3x3 layer: Wb[ci][co][kh][kw] = W[co][ci][2-kh][2-kw]."""
    key = (layer.weight.data_ptr(), layer.weight._version)
    if getattr(layer, "_bwd_key", None) != key:
        layer._bwd_packed = dense_ops.conv_pack(layer.weight.detach().flip(2, 3).transpose(0, 1).contiguous())[0]
        _publish()
        layer._bwd_key = key
    return layer._bwd_packed


def _bwd_up(layer):
    """Backward-data kernel of an up-sampling layer over the polyphase stack: input channel (a, b, co) = g_T[2y+a, 2x+b, co],
    tap (1+dy, 1+dx) carries W[co][ci][2dy+a][2dx+b] where that index exists (kh, kw <= 2)."""
    key = (layer.weight.data_ptr(), layer.weight._version)
    if getattr(layer, "_bwd_key", None) != key:
        W = layer.weight.detach()
        Co, Ci = W.shape[:2]
        wb = W.new_zeros(Ci, 2, 2, Co, 3, 3)
        for a in range(2):
            for b in range(2):
                for dy in range(2):
                    for dx in range(2):
                        kh, kw = 2 * dy + a, 2 * dx + b
                        if kh <= 2 and kw <= 2:
                            wb[:, a, b, :, 1 + dy, 1 + dx] = W[:, :, kh, kw].t()
        layer._bwd_packed = dense_ops.conv_pack(wb.reshape(Ci, 4 * Co, 3, 3).contiguous())[0]
        _publish()
        layer._bwd_key = key
    return layer._bwd_packed


def _conv_bwd_plain(layer, g_pre, s, d):
    zeros = torch.zeros(layer.in_channels, device=g_pre.device)
    return dense_ops.modulated_conv(g_pre.contiguous(), d, _bwd_plain(layer), layer.in_channels, _lib.NFE_CONV_3X3, zeros, dcoef=s,
                                    lrelu=False, act_gain=1.0, clamp=None, math=GRAD_MATH)


def _conv_bwd_up(layer, g_pre_d, s):
    """g_pre_d = d . g_pre (the demodulation factor is applied by the activation-gradient pass that produced it)."""
    N, H2, _, Co = g_pre_d.shape
    H = H2 // 2
    g_pre = g_pre_d
    # g_T = FIR^T (d . g_pre), [N,2H+1,2H+1,Co], delivered as its four polyphase images stacked along the channels [N,H+1,H+1,4Co]
    stack = dense_ops.upfirdn2d_polyphase(g_pre_d.contiguous(), padding=(2, 2), gain=4.0)
    ones = torch.ones(N, 4 * Co, device=g_pre.device)
    zeros = torch.zeros(layer.in_channels, device=g_pre.device)
    gx = dense_ops.modulated_conv(stack, ones, _bwd_up(layer), layer.in_channels, _lib.NFE_CONV_3X3, zeros, dcoef=s, lrelu=False,
                                  act_gain=1.0, clamp=None, math=GRAD_MATH)
    return gx[:, :H, :H].contiguous()


def block_forward_saving(blk, x, img, styles, dcoefs, noise_mode, conv_math):
    """One SynthesisBlock (networks_stylegan2.py:417-461, skip architecture) layer by layer -> (x_out, img_out, saved).
    styles = (conv0, conv1, torgb), dcoefs = (conv0, conv1)."""
    s0, s1, st_rgb = styles
    d0, d1 = dcoefs
    o0 = blk.conv0.forward_nhwc(x, None, noise_mode=noise_mode, conv_math=conv_math, styles=s0, dcoef=d0)
    o1 = blk.conv1.forward_nhwc(o0, None, noise_mode=noise_mode, conv_math=conv_math, styles=s1, dcoef=d1)
    y = blk.torgb.forward_nhwc(o1, None, skip=None, conv_math=conv_math, styles=st_rgb)              # clamped ToRGB output
    img = (dense_ops.upsample2d(img) if blk.conv0.up == 2 else img) + y          # networks_stylegan2.py:453-457 / superresolution.py:247-250
    return o1, img, (o0, o1, y, s0, s1, st_rgb, d0, d1)


def block_backward(blk, saved, g_img, g_x):
    """Transpose of block_forward_saving: g_img = dL/d img_out [N,r,r,3], g_x = dL/d x_out or None -> (dL/d x_in, dL/d img_in)."""
    o0, o1, y, s0, s1, st_rgb, d0, d1 = saved
    g_img = g_img.contiguous()
    clamp = blk.torgb.conv_clamp
    g_y = g_img * (y.abs() < clamp) if clamp is not None else g_img
    Wt = blk.torgb.weight.detach().reshape(blk.torgb.out_channels, blk.torgb.in_channels)
    c1, c0 = blk.conv1, blk.conv0
    # conv1's output gradient = ToRGB^T (K = 3) + what the next block sent, through conv1's activation: one pass
    g_pre1 = _act_grad(o1, g_x, c1.act_gain, c1.conv_clamp, rgb=(g_y, Wt.contiguous(), st_rgb))
    g_o0 = _conv_bwd_plain(c1, g_pre1, s1, d1)
    if c0.up == 2:
        g_pre0 = _act_grad(o0, g_o0, c0.act_gain, c0.conv_clamp, scale=d0)
        return _conv_bwd_up(c0, g_pre0, s0), dense_ops.upfirdn2d(g_img, down=2, padding=(1, 2), gain=4.0)          # transpose of upsample2d
    g_pre0 = _act_grad(o0, g_o0, c0.act_gain, c0.conv_clamp)
    return _conv_bwd_plain(c0, g_pre0, s0, d0), g_img                             # SynthesisBlockNoUp: conv0 at the same resolution, img + y


def sr_forward_saving(sr, feat, ws, noise_mode):
    """The head's forward_nhwc (two blocks, superresolution.py:29-290) layer by layer, keeping what the backward needs.  feat [N,R,R,32] NHWC."""
    assert type(sr).__name__ in HEADS, f"the SR-head gradient is not built for {type(sr).__name__}"
    ws3 = ws[:, -1:, :].repeat(1, 3, 1).to(torch.float32)                                   # superresolution.py:280
    layers = block_layers(sr.block0) + block_layers(sr.block1)
    st, dc = batch_styles(layers, ws3, [0, 1, 2, 0, 1, 2], sr.conv_math)
    # fp16 operands: the forward runs on pre-normalised weights, styles and coefficients; the backward-data convolutions use the raw
    # weights (GRAD_MATH images), so they get the raw styles and coefficients - the products styles x dcoef x weight are the same.
    st_b, dc_b = (st, dc) if dense_ops.MATH[sr.conv_math] != _lib.NFE_CONV_F16 else batch_styles(layers, ws3, [0, 1, 2, 0, 1, 2])
    x, img = feat, feat[..., :3].contiguous()
    r = sr.input_resolution
    if _resizes(sr, feat.shape[1]):                                                           # superresolution.py:283-286
        x = dense_ops.resize_bilinear(x, r, r, sr.sr_antialias)
        img = dense_ops.resize_bilinear(img, r, r, sr.sr_antialias)
    saved = [(feat.shape[1], feat.shape[2])]
    for b, blk in enumerate((sr.block0, sr.block1)):
        x, img, sv = block_forward_saving(blk, x, img, st[3 * b:3 * b + 3], (dc[3 * b], dc[3 * b + 1]), noise_mode, sr.conv_math)
        saved.append(sv[:3] + (st_b[3 * b], st_b[3 * b + 1], st_b[3 * b + 2], dc_b[3 * b], dc_b[3 * b + 1]))
    return img, saved


def sr_backward(sr, saved, g_img):
    """g_img [N,out,out,3] NHWC -> gradient w.r.t. the feature image [N,R,R,32] (its first 3 channels also feed the skip path)."""
    g_x = None                                      # gradient w.r.t. the current block's activation from the block after it
    (h_in, w_in), blocks = saved[0], saved[1:]
    for blk, sv in zip((sr.block1, sr.block0), reversed(blocks)):
        g_x, g_img = block_backward(blk, sv, g_img, g_x)
    if _resizes(sr, h_in):                          # transpose of the pre-resize of both inputs
        g_x = dense_ops.resize_bilinear_backward(g_x, h_in, w_in, sr.sr_antialias)
        g_img = dense_ops.resize_bilinear_backward(g_img.contiguous(), h_in, w_in, sr.sr_antialias)
    g_x[..., :3] += g_img                                                                         # rgb = feat[..., :3]
    return g_x


class SRImage(torch.autograd.Function):
    """image = SR(feature image); backward = sr_backward.  ws, weights and noise are constants of this function."""

    @staticmethod
    def forward(ctx, feat, sr, ws, noise_mode):
        img, saved = sr_forward_saving(sr, feat.detach(), ws.detach(), noise_mode)
        ctx.sr, ctx.saved = sr, saved
        return img

    @staticmethod
    def backward(ctx, g_img):
        return sr_backward(ctx.sr, ctx.saved, g_img), None, None, None      # saved stays: backward may run again (retain_graph)
