"""SuperresolutionHybrid8XDC with the reference's interface (training/superresolution.py:264-290): the
512^2 head train.py:277 selects.  Other SR variants (:29-153) serve other resolutions and are not built."""
import torch

from .. import dense_ops
from .networks_stylegan2 import SynthesisBlock


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
        x, rgb = self.block0.forward_nhwc(x, rgb, ws, noise_mode=noise_mode, conv_math=self.conv_math)
        x, rgb = self.block1.forward_nhwc(x, rgb, ws, noise_mode=noise_mode, conv_math=self.conv_math)
        return rgb

    def forward(self, rgb, x, ws, **block_kwargs):
        """rgb [N,3,R,R], x [N,32,R,R] NCHW -> [N,3,512,512], the reference contract."""
        out = self.forward_nhwc(dense_ops.nchw_to_nhwc(rgb.to(torch.float32)), dense_ops.nchw_to_nhwc(x.to(torch.float32)), ws,
                                **block_kwargs)
        return dense_ops.nhwc_to_nchw(out)
