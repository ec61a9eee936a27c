"""Super-resolution heads with the reference's interfaces (training/superresolution.py): SuperresolutionHybrid8XDC
(:264-290, the 512^2 head train.py:277 selects) and the variants for other output sizes / older models (:29-153):
8X (512^2, narrower), 4X and Deepfp32 (256^2), 2X (128^2), with SynthesisBlockNoUp (:158-260, 'skip' architecture)."""
import torch

from .. import dense_ops
from .networks_stylegan2 import SynthesisBlock, SynthesisLayer, ToRGBLayer, batch_styles, block_layers


def _fir_buffer():
    f = torch.tensor([1.0, 3.0, 3.0, 1.0])
    f = torch.outer(f, f)
    return f / f.sum()                                 # upfirdn2d.setup_filter([1,3,3,1])


class SynthesisBlockNoUp(torch.nn.Module):
    """superresolution.py:158-260: conv0 and conv1 at one resolution; the ToRGB output is ADDED to the incoming image
    (no upsampling, :247-250).  Only the 'skip' architecture is on this path."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture="skip",
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        super().__init__()
        assert architecture == "skip" and in_channels != 0
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16 = is_last, architecture, use_fp16
        self.register_buffer("resample_filter", _fir_buffer())
        self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_conv, self.num_torgb = 2, 1

    def forward_nhwc(self, x, img, ws, noise_mode="random", conv_math=None, pre=None, **_ignored):
        assert ws.shape[1:] == (self.num_conv + self.num_torgb, self.w_dim), f"wrong ws shape {list(ws.shape)}"
        ws = ws.to(torch.float32)
        st, dc = pre if pre is not None else batch_styles(block_layers(self), ws, range(3), conv_math)
        x = self.conv0.forward_nhwc(x, None, noise_mode=noise_mode, conv_math=conv_math, styles=st[0], dcoef=dc[0])
        x = self.conv1.forward_nhwc(x, None, noise_mode=noise_mode, conv_math=conv_math, styles=st[1], dcoef=dc[1])
        y = self.torgb.forward_nhwc(x, None, conv_math=conv_math, styles=st[2])
        return x, (img + y if img is not None else y)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, **layer_kwargs):
        img = None if img is None else dense_ops.nchw_to_nhwc(img.to(torch.float32))
        x, img = self.forward_nhwc(dense_ops.nchw_to_nhwc(x.to(torch.float32)), img, ws, **layer_kwargs)
        return dense_ops.nhwc_to_nchw(x), dense_ops.nhwc_to_nchw(img)


class _TwoBlockSR(torch.nn.Module):
    """Shared forward of the two-block heads: ws[:, -1:] x 3, optional bilinear pre-resize, block0, block1."""
    resize_if_smaller_only = False       # 4X / Deepfp32 resize only when the input is smaller (:80, :145)

    def _blocks(self, channels, c0, c1, res0, res1, noup, use_fp16, block_kwargs):
        clamp = 256 if use_fp16 else None
        B0 = SynthesisBlockNoUp if noup else SynthesisBlock
        self.block0 = B0(channels, c0, w_dim=512, resolution=res0, img_channels=3, is_last=False, use_fp16=use_fp16, conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(c0, c1, w_dim=512, resolution=res1, img_channels=3, is_last=True, use_fp16=use_fp16, conv_clamp=clamp, **block_kwargs)
        self.conv_math = None

    def forward_nhwc(self, rgb, x, ws, noise_mode="random", **_ignored):
        ws = ws[:, -1:, :].repeat(1, 3, 1)
        r = self.input_resolution
        if (x.shape[1] < r) if self.resize_if_smaller_only else (x.shape[1] != r):
            x = dense_ops.resize_bilinear(x, r, r, self.sr_antialias)
            rgb = dense_ops.resize_bilinear(rgb, r, r, self.sr_antialias)
        ws = ws.to(torch.float32)
        st, dc = batch_styles(block_layers(self.block0) + block_layers(self.block1), ws, [0, 1, 2, 0, 1, 2], self.conv_math)   # all six in one launch
        chain = hasattr(self.block0, "chains_to") and self.block0.chains_to(self.block1, ws.shape[0], self.conv_math)
        x, rgb = self.block0.forward_nhwc(x, rgb, ws, noise_mode=noise_mode, conv_math=self.conv_math, pre=(st[:3], dc[:3]),
                                          next_styles=st[3] if chain else None)
        x, rgb = self.block1.forward_nhwc(x, rgb, ws, noise_mode=noise_mode, conv_math=self.conv_math, pre=(st[3:], dc[3:]), want_x=False)
        return rgb

    def forward(self, rgb, x, ws, **block_kwargs):
        out = self.forward_nhwc(dense_ops.nchw_to_nhwc(rgb.to(torch.float32)), dense_ops.nchw_to_nhwc(x.to(torch.float32)), ws, **block_kwargs)
        return dense_ops.nhwc_to_nchw(out)


class SuperresolutionHybrid8X(_TwoBlockSR):
    """superresolution.py:29-58 (512^2 output, 128/64 channels)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        self.input_resolution, self.sr_antialias = 128, sr_antialias
        self._blocks(channels, 128, 64, 256, 512, False, sr_num_fp16_res > 0, block_kwargs)
        self.register_buffer("resample_filter", _fir_buffer())


class SuperresolutionHybrid4X(_TwoBlockSR):
    """superresolution.py:62-90 (256^2 output)."""
    resize_if_smaller_only = True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 256
        self.input_resolution, self.sr_antialias = 128, sr_antialias
        self._blocks(channels, 128, 64, 128, 256, True, sr_num_fp16_res > 0, block_kwargs)
        self.register_buffer("resample_filter", _fir_buffer())


class SuperresolutionHybrid2X(_TwoBlockSR):
    """superresolution.py:94-123 (128^2 output)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 128
        self.input_resolution, self.sr_antialias = 64, sr_antialias
        self._blocks(channels, 128, 64, 64, 128, True, sr_num_fp16_res > 0, block_kwargs)
        self.register_buffer("resample_filter", _fir_buffer())


class SuperresolutionHybridDeepfp32(_TwoBlockSR):
    """superresolution.py:127-155 (old 256^2 models; plain bilinear pre-resize, no antialias argument)."""
    resize_if_smaller_only = True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 256
        block_kwargs.pop("sr_antialias", None)      # TriPlaneGenerator always passes it; this head has no such argument
        self.input_resolution, self.sr_antialias = 128, False
        self._blocks(channels, 128, 64, 128, 256, True, sr_num_fp16_res > 0, block_kwargs)
        self.register_buffer("resample_filter", _fir_buffer())


class SuperresolutionHybrid8XDC(torch.nn.Module):
    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        use_fp16 = sr_num_fp16_res > 0
        self.input_resolution = 128
        self.sr_antialias = sr_antialias
        self.conv_math = None
        clamp = 256 if use_fp16 else None        # applied in fp32 too, as the reference does (superresolution.py:275)
        self.block0 = SynthesisBlock(channels, 256, w_dim=512, resolution=256, img_channels=3, is_last=False, use_fp16=use_fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(256, 128, w_dim=512, resolution=512, img_channels=3, is_last=True, use_fp16=use_fp16,
                                     conv_clamp=clamp, **block_kwargs)

    def forward_nhwc(self, rgb, x, ws, noise_mode="random", **_ignored):
        """rgb [N,R,R,3], x [N,R,R,32] NHWC -> image [N,512,512,3] NHWC."""
        ws = ws[:, -1:, :].repeat(1, 3, 1)                                                # superresolution.py:280
        if x.shape[1] != self.input_resolution:
            r = self.input_resolution
            x = dense_ops.resize_bilinear(x, r, r, self.sr_antialias)                     # :283-286
            rgb = dense_ops.resize_bilinear(rgb, r, r, self.sr_antialias)
        ws = ws.to(torch.float32)
        st, dc = batch_styles(block_layers(self.block0) + block_layers(self.block1), ws, [0, 1, 2, 0, 1, 2], self.conv_math)   # all six in one launch
        chain = hasattr(self.block0, "chains_to") and self.block0.chains_to(self.block1, ws.shape[0], self.conv_math)
        x, rgb = self.block0.forward_nhwc(x, rgb, ws, noise_mode=noise_mode, conv_math=self.conv_math, pre=(st[:3], dc[:3]),
                                          next_styles=st[3] if chain else None)
        x, rgb = self.block1.forward_nhwc(x, rgb, ws, noise_mode=noise_mode, conv_math=self.conv_math, pre=(st[3:], dc[3:]), want_x=False)
        return rgb

    def forward(self, rgb, x, ws, **block_kwargs):
        """rgb [N,3,R,R], x [N,32,R,R] NCHW -> [N,3,512,512], the reference contract."""
        out = self.forward_nhwc(dense_ops.nchw_to_nhwc(rgb.to(torch.float32)), dense_ops.nchw_to_nhwc(x.to(torch.float32)), ws,
                                **block_kwargs)
        return dense_ops.nhwc_to_nchw(out)
