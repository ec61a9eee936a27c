"""DisentangledImportanceRenderer with the reference's interface
(training/volumetric_rendering/renderer.py:255-363): one call into the fused HIP renderer instead of
~25 ATen ops with materialised [N,3,M*D,32] intermediates.

Jitter: the reference draws torch.rand_like / torch.rand from the global RNG (renderer.py:190,237).
Here the kernel runs Philox4x32-10 keyed by a seed drawn from torch's CPU generator (so
torch.manual_seed still makes calls reproducible); parity tests inject the uniforms through
`inject_jitter` (or the optional rendering_options keys 'jitter_coarse' / 'jitter_fine').
"""
import torch

from ... import ops


def _cross(decoder):
    """Cross-term blob of a SegmentationOSGDecoder (sigma from the other net's hidden layer); None for every other decoder."""
    fn = getattr(decoder, "packed_cross", None)
    return fn() if fn is not None else None


def generate_planes():
    """Plane axes as in renderer.py:23-37 (kept for API compatibility; the projection p0=(x,y),
    p1=(x,z), p2=(z,x) these axes define is built into the kernel)."""
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


class _RenderWithPlaneGrad(torch.autograd.Function):
    """forward = the fused render (with the sorted depths kept), backward = nfe_render_backward: what autograd does in the
    reference when `norm_planes` / `denorm_planes` are leaves being optimised (plane editing through utils.decode,
    utils.py:165-199).  Rays, jitter and decoder parameters are constants of this function."""

    @staticmethod
    def forward(ctx, norm_planes, denorm_planes, renderer, decoder, ray_origins, ray_directions, options, jitter, seed, limits):
        cross = _cross(decoder)
        noise = float(options.get("density_noise", 0) or 0)
        bf16x3 = renderer.decoder_math in (None, "bf16x3")
        if (cross is not None or noise) and not bf16x3:
            raise RuntimeError("plane gradients with SegmentationOSGDecoder or density_noise need the split-bf16 decoder (decoder_math=None): "
                               "their backward runs on the per-sample outputs the forward keeps")
        same = norm_planes is denorm_planes
        if cross is not None:       # SegmentationOSGDecoder reads the denorm features only (triplane.py:209-230): one plane set, one gradient
            pa = renderer._packed(denorm_planes.detach())
            pg, same = pa, True
        else:
            pg, pa = renderer._pack_pair(norm_planes.detach(), norm_planes.detach() if same else denorm_planes.detach())
        # with the split-bf16 decoder the forward keeps the decoders' per-sample outputs (192 B per sample) and the backward skips its
        # re-evaluation pass (nfe_render_args.tap_sample_colors, ABI v11).  With density_noise the kept sigma carries its noise, and
        # the backward - which has no draws of its own - NEEDS them; so does the two-pass backward of the cross decoder (round 6).
        colors = bf16x3 and (renderer.keep_sample_colors or noise > 0 or cross is not None)
        out = ops.render(pg, pa, decoder.packed(), options, origins=ray_origins, dirs=ray_directions, u_coarse=jitter[0],
                         u_fine=jitter[1], seed=seed, ray_limits=limits, taps=True, decoder_math=renderer.decoder_math, sample_colors=colors,
                         decoder_cross=cross, noise_values=options.get("density_noise_values"))
        ctx.save_for_backward(pg, pa, ray_origins, ray_directions, out[4]["depths_all"])
        ctx.sample_colors = out[4].get("sample_colors")
        ctx.sample_colors_resolution = out[4].get("sample_colors_resolution")
        ctx.decoder, ctx.options, ctx.same, ctx.cross = decoder, dict(options), same, cross is not None
        ctx.shape = tuple(norm_planes.shape)
        if renderer.keep_taps:
            renderer.last_taps = out[4]
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g_rgb, g_seg, g_depth, g_wsum):
        pg, pa, o, d, depths_all = ctx.saved_tensors
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        unpack = lambda g: None if g is None else g.permute(0, 1, 4, 2, 3).reshape(ctx.shape)   # gather layout -> [N,3,32,H,W]
        kw = dict(origins=o, dirs=d, sample_colors=ctx.sample_colors, sample_colors_resolution=ctx.sample_colors_resolution)
        cots = (g_rgb, g_seg, g_depth, g_wsum)
        if ctx.cross:
            # SegmentationOSGDecoder (round 6): the vector-Jacobian product is linear in the per-sample cotangents, which come from the
            # KEPT outputs of the true decoder; so it is the sum of two passes of the two-head backward over the same plane set - `net`
            # as both heads (sigma row + rgb rows, as OSGDecoder is run) and `seg_net` as the geometry head with a null appearance head.
            # norm_planes get no gradient: the reference's decoder never reads the norm features (triplane.py:209-230).
            if not need[1]:
                return (None,) * 10
            total = None
            for heads in ctx.decoder.backward_heads():
                gg, _ = ops.render_backward(pa, pa, heads, ctx.decoder.lr_mul, ctx.options, depths_all, cots, need=(True, True), **kw)
                total = gg if total is None else total + gg
            return (None, unpack(total)) + (None,) * 8
        gg, ga = ops.render_backward(pg, pg if ctx.same else pa, ctx.decoder.heads(), ctx.decoder.lr_mul, ctx.options, depths_all,
                                     cots, need=(need[0] or (ctx.same and need[1]), need[1] and not ctx.same), **kw)
        if ctx.same:        # one tensor fed both inputs: its whole gradient goes to whichever slot autograd asked for first
            g = unpack(gg)
            return (g if need[0] else None, g if (need[1] and not need[0]) else None) + (None,) * 8
        return (unpack(gg) if need[0] else None, unpack(ga) if need[1] else None) + (None,) * 8


class DisentangledImportanceRenderer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.plane_axes = generate_planes()
        self._jitter = None
        self.keep_taps = False
        self.last_taps = None
        self.decoder_math = None          # None -> split-bf16 MFMA; 'fp32' -> exact fp32 MFMA
        self.keep_sample_colors = True    # plane-gradient forward: keep the per-sample decoder outputs for the backward (1.2 GB per 4 x 128^2 x 96)
        self.seed_tensor = None
        self.cache_planes = True          # False: re-pack the NCHW plane arguments on every call (see _packed)
        self._pack_cache = []

    # -- parity hook ---------------------------------------------------------------------------------
    def inject_jitter(self, u_coarse, u_fine=None):
        """Use these uniforms ([N,M,D], [N*M,Di]) for the next forward() instead of Philox."""
        self._jitter = (u_coarse, u_fine)

    def _take_jitter(self, options):
        j, self._jitter = self._jitter, None
        if j is None and "jitter_coarse" in options:
            j = (options["jitter_coarse"], options.get("jitter_fine"))
        return j if j is not None else (None, None)

    def _seed(self):
        """Philox key of this call: a host integer drawn from torch's CPU generator, or — when `seed_tensor`
        (device int64 [1]) is set, e.g. by graphs.GraphedSynthesis — a device-resident key the kernels read."""
        if self.seed_tensor is not None:
            return self.seed_tensor
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    def _packed(self, planes):
        """Gather-layout copy of an NCHW plane tensor, cached on the tensor's identity (storage pointer, in-place version
        counter, shape): an orbit over fixed planes (utils.py:78-80 calls the renderer once per frame with the same two
        tensors) re-packs nothing; an optimiser step bumps `_version` and invalidates the entry.  Two entries: norm + denorm.
        Each entry keeps its source tensor alive, so the allocator cannot hand the same address to different data.

        The copy is published to other HIP streams by an event, not by a host sync: the entry remembers the stream that
        packed it and an event recorded behind the pack kernel; a hit on another stream (apps.StreamRing) waits for that event
        on the device.  So a plane-editing loop (one new version per optimiser step) never blocks the host here.

        Caveat (the reference re-reads the tensor on every call): writes that do not bump the version counter —
        `planes.data.add_(...)`, a view made from `.data`, a raw-pointer kernel writing in place — are invisible to the key.
        After such a write call `invalidate_plane_cache()`, or set `cache_planes = False` to pack on every call."""
        stream = torch.cuda.current_stream(planes.device)
        if not self.cache_planes:
            return ops.plane_pack(planes)
        key = (planes.data_ptr(), planes._version, tuple(planes.shape), tuple(planes.stride()), planes.device)
        for k, _, v, ev, src in self._pack_cache:
            if k == key:
                if src != stream and ev is not None:
                    stream.wait_event(ev)
                    v.record_stream(stream)       # allocated on `src`: keep the allocator from recycling it under this stream
                return v
        v = ops.plane_pack(planes)
        ev = None
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(stream)
        self._pack_cache = [(key, planes, v, ev, stream)] + self._pack_cache[:1]
        return v

    def invalidate_plane_cache(self):
        """Forget the cached gather-layout copies (see `_packed`: needed after an in-place write that bypasses autograd's
        version counter)."""
        self._pack_cache = []

    def _pack_pair(self, norm_planes, denorm_planes):
        pg = self._packed(norm_planes)
        same = (norm_planes is denorm_planes) or (norm_planes.data_ptr() == denorm_planes.data_ptr()
                                                 and norm_planes.shape == denorm_planes.shape)
        return pg, (pg if same else self._packed(denorm_planes))

    # -- reference interface -------------------------------------------------------------------------
    def forward(self, norm_planes, denorm_planes, decoder, ray_origins, ray_directions, rendering_options):
        """norm_planes/denorm_planes [N,3,32,H,W]; ray_origins/ray_directions [N,M,3] ->
        (rgb [N,M,32], seg [N,M,15], depth [N,M,1], weights.sum(2) [N,M,1])   (renderer.py:363)."""
        limits = None
        if rendering_options["ray_start"] == rendering_options["ray_end"] == "auto":      # renderer.py:312
            limits = ops.ray_limits_box(ray_origins, ray_directions, rendering_options["box_warp"])
        u_c, u_f = self._take_jitter(rendering_options)
        if torch.is_grad_enabled() and (norm_planes.requires_grad or denorm_planes.requires_grad):
            return _RenderWithPlaneGrad.apply(norm_planes, denorm_planes, self, decoder, ray_origins.detach(), ray_directions.detach(),
                                              rendering_options, (u_c, u_f), self._seed(), limits)
        pg, pa = self._pack_pair(norm_planes, denorm_planes)
        if _cross(decoder) is not None:       # SegmentationOSGDecoder: both nets read the denorm features (triplane.py:209-230)
            pg = pa
        out = ops.render(pg, pa, decoder.packed(), rendering_options, origins=ray_origins, dirs=ray_directions,
                         u_coarse=u_c, u_fine=u_f, seed=self._seed(), ray_limits=limits, taps=self.keep_taps,
                         decoder_math=self.decoder_math, decoder_cross=_cross(decoder), noise_values=rendering_options.get("density_noise_values"))
        if self.keep_taps:
            self.last_taps = out[4]
        return out[0], out[1], out[2], out[3]

    def render_raw_planes(self, packed_planes, affines, decoder, cam2world, intrinsics, resolution, rendering_options,
                          channels_first=True):
        """synthesis() fast path: raw backbone planes gathered once (single-gather identity),
        rays generated in-kernel, image-layout outputs.  Returns (rgb [N,32,M], seg [N,15,M], depth, wsum)."""
        if rendering_options["ray_start"] == rendering_options["ray_end"] == "auto":
            o, d = ops.ray_sampler(cam2world, intrinsics, resolution)
            limits = ops.ray_limits_box(o, d, rendering_options["box_warp"])
        else:
            limits = None
        u_c, u_f = self._take_jitter(rendering_options)
        out = ops.render(packed_planes, packed_planes, decoder.packed(), rendering_options, cam2world=cam2world,
                         intrinsics=intrinsics, resolution=resolution, affines=affines, u_coarse=u_c, u_fine=u_f,
                         seed=self._seed(), ray_limits=limits, channels_first=channels_first, taps=self.keep_taps,
                         decoder_math=self.decoder_math, decoder_cross=_cross(decoder))
        if self.keep_taps:
            self.last_taps = out[4]
        return out[0], out[1], out[2], out[3]

    def run_model(self, norm_planes, denorm_planes, decoder, sample_coordinates, sample_directions, options):
        """renderer.py:259-287: dict(rgb [N,P,32], sigma [N,P,1], seg [N,P,15]) at arbitrary points.
        sample_directions is unused, as in every decoder of the reference (triplane.py:249)."""
        noise = float(options.get("density_noise", 0) or 0)                       # renderer.py:285-286
        pg, pa = self._pack_pair(norm_planes, denorm_planes)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()) if noise > 0 else 0
        return ops.point_query(pg, pa, decoder.packed(), sample_coordinates, options["box_warp"],
                               decoder_math=self.decoder_math, density_noise=noise, seed=seed, decoder_cross=_cross(decoder))


class ImportanceRenderer(DisentangledImportanceRenderer):
    """The single-plane-set renderer the disentangled one derives from (renderer.py:81-167): `forward(planes, decoder,
    ray_origins, ray_directions, rendering_options)` with an `OSGDecoder` -> (rgb [N,M,32], depth [N,M,1],
    weights.sum(2) [N,M,1]).  Same fused kernel: the decoder's one MLP serves as both heads (sigma from row 0, rgb from
    rows 1..32), the segmentation head is all zeros and dropped."""

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        rgb, _, depth, wsum = super().forward(planes, planes, decoder, ray_origins, ray_directions, rendering_options)
        return rgb, depth, wsum

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        out = super().run_model(planes, planes, decoder, sample_coordinates, sample_directions, options)
        return {"rgb": out["rgb"], "sigma": out["sigma"]}
