"""StyleGAN2 generator half with the reference's class names, constructor arguments, parameter names
and call signatures (training/networks_stylegan2.py:96-555), backed by include/nfe_dense.h.

Differences from the reference that do not change results:
  * activations travel NHWC between layers; NCHW exists only at module boundaries;
  * modulation is applied to the activations and demodulation to the outputs (the reference's
    non-fused path, networks_stylegan2.py:68-77) so one weight image serves the whole batch;
  * `use_fp16` / `force_fp32` / `fused_modconv` are accepted and ignored: convolutions run on bf16 MFMA
    with fp32 operands split hi+lo (fp32-grade, default) or plain bf16 (`conv_math='bf16'`).
Inference only (no autograd through the HIP ops).
"""
import numpy as np
import os

import torch

from .. import _lib, dense_ops


def _pack_cached(module, weight, conv_math=None):
    """Packed MFMA image of a conv weight, rebuilt only when the parameter changes.  conv_math='fp16' has its own image (fp16
    operand words); the bf16 modes share one."""
    f16 = dense_ops.MATH[conv_math] == _lib.NFE_CONV_F16
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    attr = "_packed_f16" if f16 else "_packed"
    if getattr(module, attr + "_key", None) != key:
        # fp16: the reference pre-normalises the weights of its DEMODULATED layers (modulated_conv2d, :53-55); ToRGB is not demodulated
        setattr(module, attr, dense_ops.conv_pack(weight.detach(), "fp16" if f16 else None, prenormalize=f16 and isinstance(module, SynthesisLayer)))
        _publish()
        setattr(module, attr + "_key", key)
    return getattr(module, attr)


def _publish():
    """A cached device buffer is about to become visible to every later call, possibly on another HIP stream (apps.StreamRing):
    finish the kernels that fill it first.  Happens once per parameter version; skipped while a hipGraph is being captured."""
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream().synchronize()


class FullyConnectedLayer(torch.nn.Module):
    """networks_stylegan2.py:96-130."""

    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1, bias_init=0):
        super().__init__()
        assert activation in ("linear", "lrelu"), "only the activations this path uses are implemented"
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x, out=None, out_offset=0):
        return dense_ops.fully_connected(x, self.weight.detach(), None if self.bias is None else self.bias.detach(),
                                         self.weight_gain, self.bias_gain, self.activation == "lrelu", out, out_offset)

    def extra_repr(self):
        return f"in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}"


class MappingNetwork(torch.nn.Module):
    """networks_stylegan2.py:193-271."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation="lrelu", lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers, self.w_avg_beta = num_layers, w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        features_list = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        self.embed_features = embed_features
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f"fc{idx}", FullyConnectedLayer(features_list[idx], features_list[idx + 1], activation=activation,
                                                         lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer("w_avg", torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        assert not update_emas, "update_emas is a training feature"
        assert self.z_dim > 0 and self.num_ws is not None
        assert z.shape[1] == self.z_dim, f"Wrong size for dimension 1: got {z.shape[1]}, expected {self.z_dim}"   # misc.assert_shape
        N = z.shape[0]
        buf = torch.empty(N, self.z_dim + self.embed_features, device=z.device)
        dense_ops.normalize_2nd_moment(z.to(torch.float32), out=buf, out_offset=0)                 # :240
        if self.c_dim > 0:
            assert c.shape[1] == self.c_dim, f"Wrong size for dimension 1: got {c.shape[1]}, expected {self.c_dim}"
            y = self.embed(c.to(torch.float32))                                                    # :243
            dense_ops.normalize_2nd_moment(y, out=buf, out_offset=self.z_dim)
        x = buf
        for idx in range(self.num_layers):
            x = getattr(self, f"fc{idx}")(x)
        psi = float(truncation_psi)
        cutoff = self.num_ws if truncation_cutoff is None else min(int(truncation_cutoff), self.num_ws)
        return dense_ops.broadcast_truncate(x, self.w_avg if psi != 1 else None, self.num_ws, psi, cutoff)   # :257-267


class SynthesisLayer(torch.nn.Module):
    """networks_stylegan2.py:276-336."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation="lrelu", resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        assert kernel_size == 3 and up in (1, 2) and activation == "lrelu" and list(resample_filter) == [1, 3, 3, 1]
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        f = torch.tensor([1.0, 3.0, 3.0, 1.0])
        f = torch.outer(f, f)
        self.register_buffer("resample_filter", f / f.sum())          # kept for state_dict parity; the FIR is in the kernel
        self.padding = kernel_size // 2
        self.act_gain = np.sqrt(2)
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer("noise_const", torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def _strength(self):
        """noise_strength as a host float, re-read only when the parameter changes (no per-call sync)."""
        key = (self.noise_strength.data_ptr(), self.noise_strength._version)
        if getattr(self, "_ns_key", None) != key:
            self._ns, self._ns_key = float(self.noise_strength.detach()), key
        return self._ns

    def forward_nhwc(self, x, w, noise_mode="random", gain=1, conv_math=None, styles=None, next_styles=None, want_out=True, dcoef=None, rgb=None):
        """x: NHWC tensor, or a dense_ops.SplitImage made by the producing layer with this layer's `styles`.  With
        next_styles (the styles of the 3x3 layer consuming the output) returns (out, SplitImage).  styles / dcoef may
        come precomputed (batch_styles: one launch for a whole network)."""
        assert noise_mode in ["random", "const", "none"]
        in_res = self.resolution // self.up
        assert tuple(x.shape[1:]) == (in_res, in_res, self.in_channels), f"wrong input shape {list(x.shape)}"       # misc.assert_shape :314
        if styles is None:
            styles = self.affine(w)
        packed, wsq = _pack_cached(self, self.weight, conv_math)
        if dcoef is None and _is_f16(conv_math):            # fp16 operands: pre-normalised styles (networks_stylegan2.py:56) and the coefficient formed from them
            dcoef, styles = dense_ops.conv_demod(styles, wsq, prenormalize=True)
        if dcoef is None:
            dcoef = dense_ops.conv_demod(styles, wsq)
        noise, strength = None, 0.0
        if self.use_noise and noise_mode == "random":
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=styles.device)
            strength = self._strength()
        if self.use_noise and noise_mode == "const":
            noise, strength = self.noise_const, self._strength()
        mode = _lib.NFE_CONV_3X3_UP2 if self.up == 2 else _lib.NFE_CONV_3X3
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return dense_ops.modulated_conv(x, styles, packed, self.out_channels, mode, self.bias.detach(), dcoef=dcoef, noise=noise,
                                        noise_strength=strength, lrelu=True, act_gain=self.act_gain * gain, clamp=clamp,
                                        math=conv_math, next_styles=next_styles, want_out=want_out, rgb=rgb)

    def forward(self, x, w, noise_mode="random", fused_modconv=True, gain=1):
        return dense_ops.nhwc_to_nchw(self.forward_nhwc(dense_ops.nchw_to_nhwc(x), w, noise_mode=noise_mode, gain=gain))


class ToRGBLayer(torch.nn.Module):
    """networks_stylegan2.py:340-361."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        assert kernel_size == 1
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward_nhwc(self, x, w, skip=None, out_planes=False, conv_math=None, styles=None):
        """y = torgb(x) (+ upsample2d(skip), the img path of SynthesisBlock.forward :450-457)."""
        lin = self.affine
        if styles is None:
            styles = dense_ops.fully_connected(w, lin.weight.detach(), lin.bias.detach(), lin.weight_gain * self.weight_gain,
                                               lin.bias_gain * self.weight_gain)      # affine(w) * weight_gain
        packed, _ = _pack_cached(self, self.weight, conv_math)
        return dense_ops.modulated_conv(x, styles, packed, self.out_channels, _lib.NFE_CONV_1X1, self.bias.detach(), lrelu=False,
                                        act_gain=1.0, clamp=self.conv_clamp, skip=skip, out_planes=out_planes, math=conv_math)

    def forward(self, x, w, fused_modconv=True):
        return dense_ops.nhwc_to_nchw(self.forward_nhwc(dense_ops.nchw_to_nhwc(x), w))


def block_layers(block):
    """The modulated layers of a SynthesisBlock (or SynthesisBlockNoUp) in ws order."""
    return ([block.conv0] if hasattr(block, "conv0") else []) + [block.conv1, block.torgb]


def _is_f16(conv_math):
    return dense_ops.MATH[conv_math] == _lib.NFE_CONV_F16


def batch_styles(layers, ws, cols, conv_math=None):
    """Styles (and demodulation coefficients) of many layers in two launches instead of two per layer: layer i reads
    ws[:, cols[i]] (a strided column block, no copy).  -> ([styles], [dcoef or None]).  conv_math='fp16': the styles of the
    demodulated layers come back pre-normalised (each sample's row divided by its largest magnitude) and their coefficients are
    formed from the pre-normalised weights and styles, as modulated_conv2d does before an fp16 convolution (:53-66)."""
    groups = []
    for L, col in zip(layers, cols):
        lin = L.affine
        gain = L.weight_gain if isinstance(L, ToRGBLayer) else 1.0
        groups.append((ws[:, col], lin.weight.detach(), lin.bias.detach(), lin.weight_gain * gain, lin.bias_gain * gain))
    styles = dense_ops.fully_connected_grouped(groups)
    dcoefs = [None] * len(layers)
    idx = [i for i, L in enumerate(layers) if isinstance(L, SynthesisLayer)]
    if idx and _is_f16(conv_math):
        ds, sn = dense_ops.conv_demod_grouped([(styles[i], _pack_cached(layers[i], layers[i].weight, conv_math)[1]) for i in idx], prenormalize=True)
        styles = list(styles)
        for d, s_, i in zip(ds, sn, idx):
            dcoefs[i], styles[i] = d, s_
    elif idx:
        for d, i in zip(dense_ops.conv_demod_grouped([(styles[i], _pack_cached(layers[i], layers[i].weight)[1]) for i in idx]), idx):
            dcoefs[i] = d
    return styles, dcoefs


class SynthesisBlock(torch.nn.Module):
    """networks_stylegan2.py:365-465 ('skip' architecture, the only one this path instantiates)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture="skip",
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        super().__init__()
        assert architecture == "skip", "only the 'skip' architecture is on this path (networks_stylegan2.py:376)"
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16 = is_last, architecture, use_fp16
        self.fused_modconv_default = fused_modconv_default
        f = torch.tensor([1.0, 3.0, 3.0, 1.0])
        f = torch.outer(f, f)
        self.register_buffer("resample_filter", f / f.sum())
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        if in_channels != 0:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_torgb += 1

    def _fused_rgb(self, x_shape, img, st, out_planes, conv_math):
        """Arguments for evaluating this block's ToRGB inside conv1's epilogue, or None where conv1 cannot (out_planes, wide
        images, layers off the LDS-DMA path or with split-K)."""
        N, r, c1, co = x_shape[0], self.resolution, self.conv1.in_channels, self.conv1.out_channels
        if out_planes or self.img_channels > 4 or not dense_ops.fuses_rgb(_lib.NFE_CONV_3X3, conv_math, N, r, r, c1, co, self.img_channels):
            return None
        return (self.torgb.weight.detach(), st[-1], self.torgb.bias.detach(), img, self.torgb.conv_clamp)

    def chains_to(self, nxt, n, conv_math):
        """True if this block's conv1 can hand `nxt`'s conv0 its modulated bf16 input directly (conv1's epilogue writes it, the
        up-sampling conv0 takes it): no modulate-and-split pass over the fp32 activation between the blocks."""
        r = self.resolution
        if os.environ.get("NFE_NO_BLOCK_CHAIN"):          # A/B switch
            return False
        return (self.in_channels != 0 and hasattr(nxt, "conv0") and nxt.conv0.in_channels == self.conv1.out_channels
                and dense_ops.splits_in_epilogue(_lib.NFE_CONV_3X3, n, r, r, self.conv1.in_channels, self.conv1.out_channels)
                and dense_ops.can_chain(_lib.NFE_CONV_3X3_UP2, conv_math, n, r, r, nxt.conv0.in_channels, nxt.conv0.out_channels))

    def forward_nhwc(self, x, img, ws, noise_mode="random", out_planes=False, conv_math=None, pre=None, want_x=True, next_styles=None,
                     **_ignored):
        """pre: (styles, dcoefs) of this block's layers when the caller batched them for the whole network.  want_x=False
        (last block of a network): the activation itself is not returned, and not written where ToRGB runs fused.
        x may be the SplitImage the previous block made for conv0.  next_styles (only where chains_to(next block)): the styles of
        the next block's conv0; the block then returns that layer's SplitImage in place of x."""
        assert ws.shape[1:] == (self.num_conv + self.num_torgb, self.w_dim), f"wrong ws shape {list(ws.shape)}"   # :419
        ws = ws.to(torch.float32)
        st, dc = pre if pre is not None else batch_styles(block_layers(self), ws, range(ws.shape[1]), conv_math)
        xs_next = None
        if self.in_channels == 0:
            const = getattr(self, "_const_nhwc", None)
            if const is None or self._const_key != (self.const.data_ptr(), self.const._version):
                self._const_nhwc = self.const.detach().permute(1, 2, 0).contiguous()
                _publish()
                self._const_key = (self.const.data_ptr(), self.const._version)
            x = self._const_nhwc.unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
            x = self.conv1.forward_nhwc(x, None, noise_mode=noise_mode, conv_math=conv_math, styles=st[0], dcoef=dc[0])
        else:
            N, r, c1 = ws.shape[0], self.resolution, self.conv1.in_channels
            kw1 = dict(noise_mode=noise_mode, conv_math=conv_math, styles=st[1], dcoef=dc[1])
            if dense_ops.can_chain(_lib.NFE_CONV_3X3, conv_math, N, r, r, c1, self.conv1.out_channels) and c1 % 4 == 0:
                # conv0's FIR epilogue writes conv1's modulated bf16 input directly: no fp32 round trip between them
                _, xs = self.conv0.forward_nhwc(x, None, noise_mode=noise_mode, conv_math=conv_math, styles=st[0], dcoef=dc[0],
                                                next_styles=st[1], want_out=False)
                fused = self._fused_rgb(ws.shape, img, st, out_planes, conv_math)
                if fused is not None:           # conv1 + ToRGB + skip in one pass over the activation
                    if next_styles is not None:     # ... and the next block's input image: the fp32 activation is never written
                        _, img, xs_next = self.conv1.forward_nhwc(xs, None, rgb=fused, want_out=False, next_styles=next_styles, **kw1)
                        return xs_next, img
                    return self.conv1.forward_nhwc(xs, None, rgb=fused, want_out=want_x, **kw1)
                x = xs
            else:
                x = self.conv0.forward_nhwc(x, None, noise_mode=noise_mode, conv_math=conv_math, styles=st[0], dcoef=dc[0])
            if next_styles is not None:
                x, xs_next = self.conv1.forward_nhwc(x, None, next_styles=next_styles, **kw1)
            else:
                x = self.conv1.forward_nhwc(x, None, **kw1)
        img = self.torgb.forward_nhwc(x, None, skip=img, out_planes=out_planes, conv_math=conv_math, styles=st[-1])   # upsample2d(img) + y
        return (x if xs_next is None else xs_next), img

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, **layer_kwargs):
        x = None if x is None else dense_ops.nchw_to_nhwc(x.to(torch.float32))
        img = None if img is None else dense_ops.nchw_to_nhwc(img.to(torch.float32))
        x, img = self.forward_nhwc(x, img, ws, **layer_kwargs)
        return dense_ops.nhwc_to_nchw(x), dense_ops.nhwc_to_nchw(img)


class SynthesisNetwork(torch.nn.Module):
    """networks_stylegan2.py:469-525."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels, self.num_fp16_res = w_dim, img_resolution, img_channels, num_fp16_res
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels_dict = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.conv_math = None
        self.num_ws = 0
        for res in self.block_resolutions:
            in_channels = channels_dict[res // 2] if res > 4 else 0
            is_last = res == self.img_resolution
            block = SynthesisBlock(in_channels, channels_dict[res], w_dim=w_dim, resolution=res, img_channels=img_channels,
                                   is_last=is_last, **block_kwargs)
            self.num_ws += block.num_conv
            if is_last:
                self.num_ws += block.num_torgb
            setattr(self, f"b{res}", block)

    def forward_nhwc(self, ws, out_planes=False, **block_kwargs):
        """-> NHWC image [N,R,R,img_channels], or the tri-plane gather layout [N,3,R,R,32] if out_planes."""
        assert ws.shape[1:] == (self.num_ws, self.w_dim), f"wrong ws shape {list(ws.shape)}"                # :506
        block_kwargs = {k: v for k, v in block_kwargs.items() if k in ("noise_mode",)}
        ws = ws.to(torch.float32)
        blocks = [getattr(self, f"b{res}") for res in self.block_resolutions]
        layers, cols, w_idx = [], [], 0
        for block in blocks:                        # one launch for every style affine, one for every demodulation
            bl = block_layers(block)
            layers += bl
            cols += list(range(w_idx, w_idx + len(bl)))
            w_idx += block.num_conv
        st, dc = batch_styles(layers, ws, cols, self.conv_math)
        x = img = None
        w_idx = k = 0
        for i, (res, block) in enumerate(zip(self.block_resolutions, blocks)):
            n = block.num_conv + block.num_torgb
            cur = ws.narrow(1, w_idx, n)
            w_idx += block.num_conv
            chain = i + 1 < len(blocks) and block.chains_to(blocks[i + 1], ws.shape[0], self.conv_math)
            x, img = block.forward_nhwc(x, img, cur, out_planes=out_planes and res == self.img_resolution,
                                        conv_math=self.conv_math, pre=(st[k:k + n], dc[k:k + n]),
                                        next_styles=st[k + n] if chain else None, **block_kwargs)
            k += n
        return img

    def forward(self, ws, **block_kwargs):
        return dense_ops.nhwc_to_nchw(self.forward_nhwc(ws, **block_kwargs))


class Generator(torch.nn.Module):
    """networks_stylegan2.py:529-555 (the StyleGAN2Backbone of training/triplane.py:45)."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
