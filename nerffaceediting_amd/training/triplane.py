"""TriPlaneGenerator with the reference's interface (training/triplane.py:19-162) on the MI355X-native
kernels: mapping -> StyleGAN2 backbone (MFMA convs, NHWC) -> plane statistics -> fused volume render
-> super-resolution.  Drop-in for the callers listed in SURVEY.md §8(b1): same constructor, same
`mapping / synthesis / sample / sample_mixed / forward` signatures and output dict, same parameter
names (App. B), `rendering_kwargs` read at call time.
"""
import importlib

import torch

from .. import dense_ops, ops
from .networks_stylegan2 import FullyConnectedLayer, Generator as StyleGAN2Backbone, _publish
from .volumetric_rendering.ray_sampler import RaySampler
from .volumetric_rendering.renderer import DisentangledImportanceRenderer


def _construct_class_by_name(class_name, **kwargs):
    """dnnlib.util.construct_class_by_name (dnnlib/util.py:303) for 'training.superresolution.X' names."""
    mod, _, cls = class_name.rpartition(".")
    if mod.startswith("training."):
        mod = __package__.rsplit(".training", 1)[0] + "." + mod
    return getattr(importlib.import_module(mod), cls)(**kwargs)


class DisentangledOSGDecoder(torch.nn.Module):
    """training/triplane.py:232-270: geometry net (norm features -> sigma + 15 seg) and appearance net
    (denorm features -> 32 rgb).  The fused renderer reads the parameters through `packed()`; evaluation
    happens inside the render / point-query kernels."""

    def __init__(self, n_features, options):
        super().__init__()
        assert n_features == 32 and options["decoder_output_dim"] == 32 and options["decoder_seg_dim"] == 15
        self.hidden_dim = 64
        lr = options["decoder_lr_mul"]
        self.lr_mul = lr
        self.geo_net = torch.nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr), torch.nn.Softplus(),
                                           FullyConnectedLayer(self.hidden_dim, 1 + options["decoder_seg_dim"], lr_multiplier=lr))
        self.app_net = torch.nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr), torch.nn.Softplus(),
                                           FullyConnectedLayer(self.hidden_dim, options["decoder_output_dim"], lr_multiplier=lr))

    def _params(self):
        return [self.geo_net[0].weight, self.geo_net[0].bias, self.geo_net[2].weight, self.geo_net[2].bias,
                self.app_net[0].weight, self.app_net[0].bias, self.app_net[2].weight, self.app_net[2].bias]

    def heads(self):
        """The two heads as the 8 raw parameter tensors (geo w0,b0,w1,b1, app w0,b0,w1,b1), for the backward kernels."""
        return [p.detach() for p in self._params()]

    def packed(self):
        ps = self._params()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_key", None) != key:
            self._packed = ops.decoder_pack(*[p.detach() for p in ps], lr_mul=self.lr_mul)
            _publish()
            self._packed_key = key
        return self._packed

    decoder_math = None        # None -> split-bf16 MFMA; 'fp32' -> exact fp32 MFMA (as DisentangledImportanceRenderer.decoder_math)

    def forward(self, sampled_norm_features, sampled_denorm_features, ray_directions):
        """triplane.py:249-270: features [N,3,M,32] (x2) -> dict(rgb [N,M,32], sigma [N,M,1], seg [N,M,15]).  The renderer
        never calls this (the decoder is evaluated inside the fused kernels); it serves callers that sample features themselves."""
        return ops.decoder_forward(sampled_norm_features, sampled_denorm_features, self.packed(), decoder_math=self.decoder_math)


class OSGDecoder(torch.nn.Module):
    """training/triplane.py:167-190 (the EG3D decoder: one MLP 32 -> 64 -> 1 + 32, no segmentation).  For the fused kernels
    it is presented as two heads over the same hidden layer: geometry = output row 0 (sigma; seg rows zero), appearance
    = rows 1..32 - the split the reference applies when it resumes from such a pickle (training_loop.py:202-214)."""

    def __init__(self, n_features, options):
        super().__init__()
        assert n_features == 32 and options["decoder_output_dim"] == 32
        self.hidden_dim = 64
        lr = options["decoder_lr_mul"]
        self.lr_mul = lr
        self.net = torch.nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr), torch.nn.Softplus(),
                                       FullyConnectedLayer(self.hidden_dim, 1 + options["decoder_output_dim"], lr_multiplier=lr))

    def heads(self):
        w0, b0, w2, b2 = (p.detach() for p in (self.net[0].weight, self.net[0].bias, self.net[2].weight, self.net[2].bias))
        gw = torch.zeros(16, 64, device=w2.device); gb = torch.zeros(16, device=w2.device)
        gw[:1], gb[:1] = w2[:1], b2[:1]
        return [w0, b0, gw, gb, w0, b0, w2[1:].contiguous(), b2[1:].contiguous()]

    def packed(self):
        ps = [self.net[0].weight, self.net[0].bias, self.net[2].weight, self.net[2].bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_key", None) != key:
            self._packed = ops.decoder_pack(*self.heads(), lr_mul=self.lr_mul)
            _publish()
            self._packed_key = key
        return self._packed

    decoder_math = None

    def forward(self, sampled_features, ray_directions):
        """triplane.py:178-190: features [N,3,M,32] -> dict(rgb [N,M,32], sigma [N,M,1])."""
        out = ops.decoder_forward(sampled_features, sampled_features, self.packed(), decoder_math=self.decoder_math)
        return {"rgb": out["rgb"], "sigma": out["sigma"]}


class SegmentationOSGDecoder(torch.nn.Module):
    """training/triplane.py:192-230 (the `disable_alignment` ablation): `net` 32 -> 64 -> 1 + 32 gives sigma and rgb, `seg_net`
    32 -> 64 -> 15 the segmentation, both from the SAME (denorm) features.  For the fused kernels `seg_net` is the geometry
    head (sigma row zero), `net` rows 1..32 the appearance head and `net` row 0 a cross term from the appearance head's hidden
    layer into the sigma output (nfe_decoder_pack_cross)."""

    def __init__(self, n_features, options):
        super().__init__()
        assert n_features == 32 and options["decoder_output_dim"] == 32 and options["decoder_seg_dim"] == 15
        self.hidden_dim = 64
        lr = options["decoder_lr_mul"]
        self.lr_mul = lr
        self.net = torch.nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr), torch.nn.Softplus(),
                                       FullyConnectedLayer(self.hidden_dim, 1 + options["decoder_output_dim"], lr_multiplier=lr))
        self.seg_net = torch.nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr), torch.nn.Softplus(),
                                           FullyConnectedLayer(self.hidden_dim, options["decoder_seg_dim"], lr_multiplier=lr))

    def _params(self):
        return [self.net[0].weight, self.net[0].bias, self.net[2].weight, self.net[2].bias,
                self.seg_net[0].weight, self.seg_net[0].bias, self.seg_net[2].weight, self.seg_net[2].bias]

    def heads(self):
        nw0, nb0, nw2, nb2, sw0, sb0, sw2, sb2 = (p.detach() for p in self._params())
        gw = torch.zeros(16, 64, device=nw2.device); gb = torch.zeros(16, device=nw2.device)
        gw[1:], gb[1:], gb[:1] = sw2, sb2, nb2[:1]
        return [sw0, sb0, gw, gb, nw0, nb0, nw2[1:].contiguous(), nb2[1:].contiguous()]

    def backward_heads(self):
        """The two parameter sets whose two-head backward passes add up to this decoder's plane gradient (renderer.py,
        _RenderWithPlaneGrad.backward): `net` as both heads (sigma row + rgb rows), and `seg_net` as the geometry head (sigma row
        zero) beside a null appearance head."""
        nw0, nb0, nw2, nb2, sw0, sb0, sw2, sb2 = (p.detach() for p in self._params())
        ga = torch.zeros(16, 64, device=nw2.device); gab = torch.zeros(16, device=nw2.device)
        ga[:1], gab[:1] = nw2[:1], nb2[:1]
        gb = torch.zeros(16, 64, device=nw2.device); gbb = torch.zeros(16, device=nw2.device)
        gb[1:], gbb[1:] = sw2, sb2
        return ([nw0, nb0, ga, gab, nw0, nb0, nw2[1:].contiguous(), nb2[1:].contiguous()],
                [sw0, sb0, gb, gbb, nw0, nb0, torch.zeros_like(nw2[1:]).contiguous(), torch.zeros_like(nb2[1:]).contiguous()])

    def cross(self):
        x = torch.zeros(16, 64, device=self.net[2].weight.device)
        x[:1] = self.net[2].weight.detach()[:1]
        return x

    def _refresh(self):
        key = tuple((p.data_ptr(), p._version) for p in self._params())
        if getattr(self, "_packed_key", None) != key:
            self._packed = ops.decoder_pack(*self.heads(), lr_mul=self.lr_mul)
            self._packed_cross = ops.decoder_pack_cross(self.cross(), lr_mul=self.lr_mul)
            _publish()
            self._packed_key = key

    def packed(self):
        self._refresh()
        return self._packed

    def packed_cross(self):
        self._refresh()
        return self._packed_cross

    def forward(self, sampled_norm_features, sampled_denorm_features, ray_directions):
        """triplane.py:209-230: both nets read the DENORM features (the norm features are ignored, as in the reference)."""
        return ops.decoder_forward(sampled_denorm_features, sampled_denorm_features, self.packed(), decoder_cross=self.packed_cross())


def packed_cross_of(decoder):
    """The cross-term blob of a decoder, or None (every decoder but SegmentationOSGDecoder)."""
    fn = getattr(decoder, "packed_cross", None)
    return fn() if fn is not None else None


class TriPlaneGenerator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={},
                 rendering_kwargs={}, sr_kwargs={}, disable_disentangle=False, disable_alignment=False, **synthesis_kwargs):
        super().__init__()
        assert not disable_alignment or disable_disentangle                                   # triplane.py:42
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        # disable_disentangle (triplane.py:93,104-107,119): no normalisation, both decoder heads read the raw planes
        self.disable_disentangle, self.disable_alignment = bool(disable_disentangle), bool(disable_alignment)
        self.init_args, self.init_kwargs = (), dict(                      # what persistence.persistent_class records
            z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
            sr_num_fp16_res=sr_num_fp16_res, mapping_kwargs=mapping_kwargs, rendering_kwargs=rendering_kwargs,
            sr_kwargs=sr_kwargs, disable_disentangle=bool(disable_disentangle), disable_alignment=bool(disable_alignment), **synthesis_kwargs)
        self.renderer = DisentangledImportanceRenderer()
        self.ray_sampler = RaySampler()
        self.backbone = StyleGAN2Backbone(z_dim, c_dim, w_dim, img_resolution=256, img_channels=32 * 3,
                                          mapping_kwargs=mapping_kwargs, **synthesis_kwargs)
        self.superresolution = _construct_class_by_name(class_name=rendering_kwargs["superresolution_module"], channels=32,
                                                        img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res,
                                                        sr_antialias=rendering_kwargs["sr_antialias"], **sr_kwargs)
        decoder_class = SegmentationOSGDecoder if disable_alignment else DisentangledOSGDecoder       # triplane.py:48-51
        self.decoder = decoder_class(32, {"decoder_lr_mul": rendering_kwargs.get("decoder_lr_mul", 1),
                                          "decoder_output_dim": 32, "decoder_seg_dim": 15})
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self._last_planes = None

    # ---- plane statistics (triplane.py:56-68), NCHW in / out as the reference's helpers ------------
    def compute_mean_var(self, planes):
        return ops.plane_stats(planes)

    def normalize_plane(self, planes):
        mean, var = ops.plane_stats(planes)
        gs, gb, _, _ = ops.make_affine(mean, var)
        N, C = planes.shape[:2]
        return ops.plane_affine(planes, gs.reshape(N, C, 1, 1), gb.reshape(N, C, 1, 1)), mean, var

    def denormalize_plane(self, planes, mean, var):
        return ops.plane_affine(planes, var, mean)

    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs["c_gen_conditioning_zero"]:
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get("c_scale", 0), truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    def _planes(self, ws, synthesis_kwargs):
        """backbone -> (tri-plane gather layout [N,3,H,W,32], mean, std [N,96,1,1])."""
        packed = self.backbone.synthesis.forward_nhwc(ws, out_planes=True, **synthesis_kwargs)
        N, _, H, W, _ = packed.shape
        mean, std = dense_ops.plane_stats_nhwc(packed.view(N * 3, H, W, 32))      # channel p*32+c <-> row (n*3+p), col c
        return packed, mean.reshape(N, 96, 1, 1), std.reshape(N, 96, 1, 1)

    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                  use_cached_backbone=False, planes_mean=None, planes_var=None, **synthesis_kwargs):
        cam2world_matrix = c[:, :16].reshape(-1, 4, 4).to(torch.float32)
        intrinsics = c[:, 16:25].reshape(-1, 3, 3).to(torch.float32)
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        N = cam2world_matrix.shape[0]
        R = neural_rendering_resolution

        if use_cached_backbone and self._last_planes is not None:
            packed, mean, var = self._cached_planes()
        else:
            packed, mean, var = self._planes(ws, synthesis_kwargs)
        stage_events = getattr(self, "stage_events", None)      # bench.py: [.., after backbone, after render, ..]
        if stage_events is not None:
            stage_events[1].record()

        # normalisation + optional appearance override as per-channel affines on the sampled values
        # (single-gather identity; triplane.py:93-103 incl. the (int,int) special case)
        new_mean = new_var = None
        if planes_mean is not None and planes_var is not None:
            if type(planes_mean) == int and type(planes_var) == int:
                new_mean, new_var = mean[planes_mean][None].contiguous(), var[planes_var][None].contiguous()
            else:
                new_mean, new_var = planes_mean, planes_var
        affines = ops.make_affine(mean, var, new_mean, new_var)
        if cache_backbone:
            self._store_planes(packed, mean, var, affines if (new_mean is not None and not self.disable_disentangle) else None)
        if self.disable_disentangle:                # identity affines: sampled raw values go to both heads; no statistics returned
            affines = tuple(torch.ones_like(a) if i % 2 == 0 else torch.zeros_like(a) for i, a in enumerate(affines))
            mean = var = None

        feature_samples, seg_samples, depth_samples, _ = self.renderer.render_raw_planes(
            packed, affines, self.decoder, cam2world_matrix, intrinsics, R, self.rendering_kwargs, channels_first=False)

        if stage_events is not None:
            stage_events[2].record()
        # [N,M,32] is already NHWC for the SR head; the reference's NCHW images are produced for the output dict
        feat_nhwc = feature_samples.view(N, R, R, 32)
        rgb_nhwc = feat_nhwc[..., :3].contiguous()
        sr_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != "noise_mode"}
        sr_nhwc = self.superresolution.forward_nhwc(rgb_nhwc, feat_nhwc, ws,
                                                    noise_mode=self.rendering_kwargs["superresolution_noise_mode"], **sr_kwargs)
        return {
            "image": dense_ops.nhwc_to_nchw(sr_nhwc),
            "image_seg": dense_ops.nhwc_to_nchw(seg_samples.view(N, R, R, 15)),
            "image_raw": dense_ops.nhwc_to_nchw(rgb_nhwc),
            "image_depth": depth_samples.permute(0, 2, 1).reshape(N, 1, R, R),
            "plane_mean": mean,
            "plane_var": var,
        }

    # ---- backbone cache (triplane.py:88-89,109-110) -------------------------------------------------
    # `_last_planes` is what the reference stores: the NCHW planes [N,96,H,W] AFTER an appearance override (so a later
    # use_cached_backbone=True call keeps the overridden appearance and reports its statistics).  The gather-layout copy
    # and its statistics live beside it, keyed on the tensor's identity: a caller that assigns its own tensor to
    # `_last_planes` gets it re-packed on first use.
    def _store_planes(self, packed, mean, var, override_affines):
        N, _, H, W, _ = packed.shape
        nchw = dense_ops.nhwc_to_nchw(packed.view(N * 3, H, W, 32)).view(N, 96, H, W)
        if override_affines is not None:            # planes = denormalize_plane(norm_planes, mean', var'), triplane.py:98-103
            _, _, a_scale, a_shift = override_affines
            nchw = ops.plane_affine(nchw, a_scale.reshape(N, 96, 1, 1), a_shift.reshape(N, 96, 1, 1))
            packed = ops.plane_pack(nchw)
            mean, var = ops.plane_stats(nchw)
        self._last_planes = nchw
        self._last_packed = ((nchw.data_ptr(), nchw._version, tuple(nchw.shape)), packed, mean, var)

    def invalidate_cached_planes(self):
        """Drop the gather-layout copy of `_last_planes` (and the renderer's pack cache).  The copies are keyed on (storage
        pointer, autograd version counter, shape); an in-place write that bypasses the counter (`.data`, a raw-pointer kernel)
        needs this call before the next use_cached_backbone=True / renderer call — the reference re-reads the tensor each time."""
        self._last_packed = None
        self.renderer.invalidate_plane_cache()

    def _cached_planes(self):
        p = self._last_planes
        key = (p.data_ptr(), p._version, tuple(p.shape))
        cached = getattr(self, "_last_packed", None)
        if cached is None or cached[0] != key:
            p4 = p.reshape(p.shape[0], 96, p.shape[-2], p.shape[-1]).to(torch.float32)
            mean, var = ops.plane_stats(p4)
            cached = self._last_packed = (key, ops.plane_pack(p4), mean, var)
        return cached[1], cached[2], cached[3]

    def _sample_planes(self, ws, coordinates, synthesis_kwargs):
        packed, mean, var = self._planes(ws, synthesis_kwargs)
        affines = ops.make_affine(mean, var)
        if self.disable_disentangle:
            affines = tuple(torch.ones_like(a) if i % 2 == 0 else torch.zeros_like(a) for i, a in enumerate(affines))
        # run_model receives self.rendering_kwargs (triplane.py:148,157), so density_noise applies here too (renderer.py:285-286)
        noise = float(self.rendering_kwargs.get("density_noise", 0) or 0)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()) if noise > 0 else 0
        return ops.point_query(packed, packed, self.decoder.packed(), coordinates.to(torch.float32),
                               self.rendering_kwargs["box_warp"], affines=affines, density_noise=noise, seed=seed,
                               decoder_math=self.renderer.decoder_math, decoder_cross=packed_cross_of(self.decoder))

    def sample(self, coordinates, directions, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self._sample_planes(ws, coordinates, synthesis_kwargs)

    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        return self._sample_planes(ws, coordinates, synthesis_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, planes_mean=None, planes_var=None, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, planes_mean=planes_mean,
                              planes_var=planes_var, **synthesis_kwargs)
