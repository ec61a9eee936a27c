"""Tensor-level wrappers over the C ABI: validate, allocate outputs with torch, pass raw pointers and
the current HIP stream.  PyTorch is plumbing here (memory + stream); all arithmetic is in the library."""
import ctypes

import torch

from . import _lib

_workspaces = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: tensor is on {t.device}; this path runs only on the GPU (no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if shape is not None:
        assert len(shape) == t.dim() and all(s is None or s == d for s, d in zip(shape, t.shape)), \
            f"{name}: wrong shape {list(t.shape)}, expected {list(shape)}"      # misc.assert_shape convention
    return t.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _math_mode(mode):
    """None/'bf16x3' -> split-bf16 MFMA decoder (default); 'fp32' -> exact fp32 MFMA."""
    if mode in (None, "bf16x3", _lib.NFE_MATH_BF16X3):
        return _lib.NFE_MATH_BF16X3
    if mode in ("fp32", _lib.NFE_MATH_FP32):
        return _lib.NFE_MATH_FP32
    raise ValueError(f"unknown decoder_math {mode!r}")


def _workspace(device, nbytes):
    if torch.cuda.is_current_stream_capturing():          # graph capture: memory must come from the graph's pool
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _call_ws(device):
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return device, _workspaces.get((device.index, torch.cuda.current_stream(device).cuda_stream))


def render_call_status(device=None, backward=False):
    """The per-call error surface of a lost wave hand-off (nfe_render_call_status / nfe_render_backward_call_status, ABI v15):
    synchronises the current stream of `device` and raises RuntimeError (NFE_EHANDOFF, -4) when the LAST ops.render (backward=True:
    ops.render_backward) call issued from that stream abandoned a producer / consumer wait - every output of that call is NaN then.
    Returns None otherwise.  This is the check to put at one's own synchronisation point; no later call is refused because of an
    earlier call's failure."""
    device, ws = _call_ws(device)
    if ws is None:
        return
    lib = _lib.load()
    fn = lib.nfe_render_backward_call_status if backward else lib.nfe_render_call_status
    with torch.cuda.device(device):
        _lib.check(fn(ctypes.c_void_p(ws.data_ptr()), _stream(), None), fn.__name__)


def render_handoff_aborts(device=None, backward=False):
    """Diagnostic (synchronises): how many producer / consumer waits the wave-specialised kernel abandoned in the LAST ops.render
    (backward=True: ops.render_backward) call issued from the current stream of `device` - the count nfe_render_call_status reads.
    Must be 0: a non-zero count means a wave pair lost its partner, finished with garbage instead of hanging the GPU, and every output
    of that call was overwritten with NaN."""
    device, ws = _call_ws(device)
    if ws is None:
        return 0
    lib = _lib.load()
    lost = ctypes.c_uint32(0)
    fn = lib.nfe_render_backward_call_status if backward else lib.nfe_render_call_status
    with torch.cuda.device(device):
        rc = fn(ctypes.c_void_p(ws.data_ptr()), _stream(), ctypes.byref(lost))
    if rc not in (0, -4):
        _lib.check(rc, fn.__name__)
    return int(lost.value)


def render_status(clear=False):
    """nfe_render_status (no synchronisation): (lost_handoffs, poisoned_calls) of this PROCESS since the last clear - a sticky
    diagnostic fed by every stream and thread.  A render or render-backward call whose wave-specialised kernel abandoned a hand-off
    wait has NaN in all its outputs and is counted here once its closing kernel has run.  No call is refused because of it;
    raise_if_handoff_lost() turns it into an exception at a point where the caller has synchronised anyway."""
    lost, calls = ctypes.c_uint32(0), ctypes.c_uint32(0)
    _lib.check(_lib.load().nfe_render_status(ctypes.byref(lost), ctypes.byref(calls), 1 if clear else 0), "nfe_render_status")
    return int(lost.value), int(calls.value)


def raise_if_handoff_lost():
    """For the caller's own synchronisation points (frames copied to the host, a loss value read): raise RuntimeError when any render
    / backward call of this process has been poisoned since the last clear (and clear).  No synchronisation of its own."""
    lost, calls = render_status()
    if lost:
        lost, calls = render_status(clear=True)
        raise RuntimeError(f"{calls} render call(s) of this process lost {lost} wave hand-offs (NFE_EHANDOFF): their outputs are NaN; "
                           "repeat them (NFE_RENDER_WS=0 / NFE_BWD_DECODER=single select the kernels without a hand-off)")


def render_last_kernels():
    """Names of the render kernels the last ops.render call of this thread launched (nfe_render_last_kernels), as a list."""
    s = _lib.load().nfe_render_last_kernels()
    return s.decode().split() if s else []


def ray_sampler(cam2world, intrinsics, resolution):
    """RaySampler.forward (ray_sampler.py:24-62): [N,4,4],[N,3,3] -> origins, dirs [N,R*R,3]."""
    lib = _lib.load()
    cam2world = _dev(cam2world, "cam2world_matrix", (None, 4, 4))
    N = cam2world.shape[0]
    intrinsics = _dev(intrinsics, "intrinsics", (N, 3, 3))
    R = int(resolution)
    o = torch.empty(N, R * R, 3, device=cam2world.device)
    d = torch.empty_like(o)
    with torch.cuda.device(cam2world.device):
        _lib.check(lib.nfe_ray_sampler(_ptr(cam2world), _ptr(intrinsics), N, R, _ptr(o), _ptr(d), _stream()), "nfe_ray_sampler")
    return o, d


def ray_limits_box(origins, dirs, box_side_length):
    """get_ray_limits_box + the invalid-ray fix-up of renderer.py:312-318 -> (ray_start, ray_end) [N,M,1]."""
    lib = _lib.load()
    origins = _dev(origins, "ray_origins", (None, None, 3))
    dirs = _dev(dirs, "ray_directions", tuple(origins.shape))
    N, M = origins.shape[:2]
    rs = torch.empty(N, M, 1, device=origins.device)
    re = torch.empty_like(rs)
    scratch = torch.empty(2, dtype=torch.int32, device=origins.device)
    with torch.cuda.device(origins.device):
        _lib.check(lib.nfe_ray_limits_box(_ptr(origins), _ptr(dirs), N * M, float(box_side_length), _ptr(rs), _ptr(re),
                                          _ptr(scratch), _stream()), "nfe_ray_limits_box")
    return rs, re


def plane_stats(planes):
    """compute_mean_var (triplane.py:56-60): [N,C,H,W] -> mean, std [N,C,1,1]."""
    lib = _lib.load()
    planes = _dev(planes, "planes", (None, None, None, None))
    N, C, H, W = planes.shape
    mean = torch.empty(N, C, 1, 1, device=planes.device)
    std = torch.empty_like(mean)
    with torch.cuda.device(planes.device):
        _lib.check(lib.nfe_plane_stats(_ptr(planes), N, C, H * W, _ptr(mean), _ptr(std), _stream()), "nfe_plane_stats")
    return mean, std


def plane_affine(planes, scale, shift):
    """out = planes*scale + shift with scale/shift [N|1,C,1,1] (normalize/denormalize_plane)."""
    lib = _lib.load()
    planes = _dev(planes, "planes", (None, None, None, None))
    N, C, H, W = planes.shape
    scale = _dev(scale, "scale").reshape(-1, C)
    shift = _dev(shift, "shift").reshape(-1, C)
    na = scale.shape[0]
    assert shift.shape[0] == na and na in (1, N), "scale/shift must be [N,C,1,1] or [1,C,1,1]"
    out = torch.empty_like(planes)
    with torch.cuda.device(planes.device):
        _lib.check(lib.nfe_plane_affine(_ptr(planes), _ptr(scale), _ptr(shift), N, C, H * W, na, _ptr(out), _stream()),
                   "nfe_plane_affine")
    return out


def make_affine(mean, std, new_mean=None, new_std=None):
    """Affines for the single-gather identity (include/nfe_render.h: nfe_make_affine). Returns 4x [N,C]."""
    lib = _lib.load()
    mean = _dev(mean, "mean")
    N, C = mean.shape[0], mean.shape[1]
    mean = mean.reshape(N, C)
    std = _dev(std, "std").reshape(N, C)
    no = 0
    if new_mean is not None:
        new_mean = _dev(new_mean, "planes_mean").reshape(-1, C)
        new_std = _dev(new_std, "planes_var").reshape(-1, C)
        no = new_mean.shape[0]
        assert new_std.shape[0] == no and no in (1, N), "override statistics must be [N,C,1,1] or [1,C,1,1]"
    outs = [torch.empty(N, C, device=mean.device) for _ in range(4)]
    with torch.cuda.device(mean.device):
        _lib.check(lib.nfe_make_affine(_ptr(mean), _ptr(std), _ptr(new_mean), _ptr(new_std), N, C, no,
                                       *[_ptr(o) for o in outs], _stream()), "nfe_make_affine")
    return outs


def plane_pack(planes):
    """[N,96,H,W] or [N,3,32,H,W] -> gather layout [N,3,H,W,32]."""
    lib = _lib.load()
    if planes.dim() == 5:
        planes = planes.reshape(planes.shape[0], 96, planes.shape[-2], planes.shape[-1])
    planes = _dev(planes, "planes", (None, 96, None, None))
    N, _, H, W = planes.shape
    out = torch.empty(N, 3, H, W, 32, device=planes.device)
    with torch.cuda.device(planes.device):
        _lib.check(lib.nfe_plane_pack(_ptr(planes), N, H, W, _ptr(out), _stream()), "nfe_plane_pack")
    return out


def decoder_pack(geo_w0, geo_b0, geo_w1, geo_b1, app_w0, app_b0, app_w1, app_b1, lr_mul=1.0):
    lib = _lib.load()
    ts = [_dev(geo_w0, "geo_net.0.weight", (64, 32)), _dev(geo_b0, "geo_net.0.bias", (64,)),
          _dev(geo_w1, "geo_net.2.weight", (16, 64)), _dev(geo_b1, "geo_net.2.bias", (16,)),
          _dev(app_w0, "app_net.0.weight", (64, 32)), _dev(app_b0, "app_net.0.bias", (64,)),
          _dev(app_w1, "app_net.2.weight", (32, 64)), _dev(app_b1, "app_net.2.bias", (32,))]
    out = torch.empty(_lib.NFE_DECODER_PACKED_FLOATS, device=ts[0].device)
    with torch.cuda.device(out.device):
        _lib.check(lib.nfe_decoder_pack(*[_ptr(t) for t in ts], float(lr_mul), _ptr(out), _stream()), "nfe_decoder_pack")
    return out


def decoder_pack_cross(cross_w1, lr_mul=1.0):
    """nfe_decoder_pack_cross: [16,64] geometry-head rows on the appearance head's hidden units (SegmentationOSGDecoder)."""
    lib = _lib.load()
    cross_w1 = _dev(cross_w1, "cross_w1", (16, 64))
    out = torch.empty(_lib.NFE_DECODER_CROSS_FLOATS, device=cross_w1.device)
    with torch.cuda.device(out.device):
        _lib.check(lib.nfe_decoder_pack_cross(_ptr(cross_w1), float(lr_mul), _ptr(out), _stream()), "nfe_decoder_pack_cross")
    return out


def decoder_forward(features_geo, features_app, decoder_packed, decoder_math=None, decoder_cross=None):
    """nfe_decoder_forward: sampled features [N,n_planes,P,32] (x2; the same tensor twice for a single-set decoder) ->
    dict(rgb [N,P,32], sigma [N,P,1], seg [N,P,15]) — the decoders' own forward() (triplane.py:178-190, 209-230, 249-270)."""
    lib = _lib.load()
    features_geo = _dev(features_geo, "sampled_norm_features", (None, None, None, 32))
    features_app = features_geo if features_app is features_geo else _dev(features_app, "sampled_denorm_features", tuple(features_geo.shape))
    N, n_planes, P, _ = features_geo.shape
    dev = features_geo.device
    decoder_packed = _dev(decoder_packed, "decoder_packed", (_lib.NFE_DECODER_PACKED_FLOATS,))
    if decoder_cross is not None:
        decoder_cross = _dev(decoder_cross, "decoder_cross", (_lib.NFE_DECODER_CROSS_FLOATS,))
    rgb = torch.empty(N, P, 32, device=dev)
    sigma = torch.empty(N, P, 1, device=dev)
    seg = torch.empty(N, P, 15, device=dev)
    if N * P > 0:
        with torch.cuda.device(dev):
            _lib.check(lib.nfe_decoder_forward(_ptr(features_geo), _ptr(features_app), N, n_planes, P, _ptr(decoder_packed),
                                               _math_mode(decoder_math), _ptr(decoder_cross), _ptr(rgb), _ptr(sigma), _ptr(seg),
                                               _stream()), "nfe_decoder_forward")
    return {"rgb": rgb, "sigma": sigma, "seg": seg}


def render(planes_geo, planes_app, decoder_packed, options, *, origins=None, dirs=None, cam2world=None,
           intrinsics=None, resolution=0, affines=None, u_coarse=None, u_fine=None, seed=0,
           channels_first=False, taps=False, ray_limits=None, decoder_math=None, decoder_cross=None, clock_probe=None,
           sample_colors=False, noise_values=None):
    """nfe_render.  planes_* are packed [Np,3,H,W,32] (Np == N or 1); affines = 4x [N,96] or None.
    clock_probe: optional int64 device tensor [4] the final render launch stamps (nfe_render_args.clock_probe).
    noise_values: optional [N,M,D+Di] standard normals for options['density_noise'] instead of the Philox draws (parity hook:
    coarse sample k at k, the fine sample of ascending rank r at D + r; nfe_render_args.density_noise_values).

    Returns (rgb, seg, depth, wsum[, taps]) with rgb [N,M,32] (or [N,32,M] if channels_first),
    seg [N,M,15], depth [N,M,1], wsum [N,M,1] — the tuple DisentangledImportanceRenderer.forward
    returns (renderer.py:363).
    """
    lib = _lib.load()
    planes_geo = _dev(planes_geo, "planes_geo", (None, 3, None, None, 32))
    planes_app = planes_geo if planes_app is planes_geo else _dev(planes_app, "planes_app", tuple(planes_geo.shape))
    Np, _, H, W, _ = planes_geo.shape
    dev = planes_geo.device
    if origins is not None:
        origins = _dev(origins, "ray_origins", (None, None, 3))
        N, M = origins.shape[0], origins.shape[1]
        dirs = _dev(dirs, "ray_directions", (N, M, 3))
    else:
        cam2world = _dev(cam2world, "cam2world_matrix", (None, 4, 4))
        N = cam2world.shape[0]
        intrinsics = _dev(intrinsics, "intrinsics", (N, 3, 3))
        M = int(resolution) ** 2
    assert Np in (1, N), f"planes batch {Np} must be 1 or equal the ray batch {N}"
    D = int(options["depth_resolution"])
    Di = int(options.get("depth_resolution_importance", 0) or 0)
    if options.get("clamp_mode", "softplus") != "softplus":
        raise AssertionError("MipRayMarcher only supports `clamp_mode`=`softplus`!")      # ray_marcher.py:78
    a = _lib.RenderArgs()
    a.struct_size = ctypes.sizeof(_lib.RenderArgs)
    a.density_noise = float(options.get("density_noise", 0) or 0)          # renderer.py:285-286 (Philox normals, see header)
    a.planes_geo, a.planes_app = planes_geo.data_ptr(), planes_app.data_ptr()
    a.plane_h, a.plane_w = H, W
    a.plane_view_stride = 0 if (Np == 1 and N > 1) else 3 * H * W * 32
    keep = [planes_geo, planes_app, decoder_packed, origins, dirs, cam2world, intrinsics]
    if affines is not None:
        affines = [_dev(t, "affine", (N, 96)) for t in affines]
        a.geo_scale, a.geo_shift, a.app_scale, a.app_shift = [t.data_ptr() for t in affines]
        keep += affines
    a.decoder_packed = _dev(decoder_packed, "decoder_packed", (_lib.NFE_DECODER_PACKED_FLOATS,)).data_ptr()
    a.decoder_math = _math_mode(decoder_math)
    if decoder_cross is not None:
        decoder_cross = _dev(decoder_cross, "decoder_cross", (_lib.NFE_DECODER_CROSS_FLOATS,))
        a.decoder_cross = decoder_cross.data_ptr()
        keep.append(decoder_cross)
    if noise_values is not None:
        noise_values = _dev(noise_values, "noise_values", (N, M, D + Di))
        a.density_noise_values = noise_values.data_ptr()
        keep.append(noise_values)
    if clock_probe is not None:
        assert clock_probe.is_cuda and clock_probe.dtype == torch.int64 and clock_probe.numel() >= 4
        a.clock_probe = clock_probe.data_ptr()
        keep.append(clock_probe)
    a.n_views, a.n_rays = N, M
    if origins is not None:
        a.origins, a.dirs = origins.data_ptr(), dirs.data_ptr()
        r = int(round(M ** 0.5))
        a.resolution = r if (resolution == 0 and r * r == M) else int(resolution)
    else:
        a.cam2world, a.intrinsics, a.resolution = cam2world.data_ptr(), intrinsics.data_ptr(), int(resolution)
    a.depth_resolution, a.depth_resolution_importance = D, Di
    if ray_limits is not None:
        rs, re = (_dev(t, "ray_limits").reshape(N, M) for t in ray_limits)
        a.ray_start_per_ray, a.ray_end_per_ray = rs.data_ptr(), re.data_ptr()
        keep += [rs, re]
    else:
        a.ray_start, a.ray_end = float(options["ray_start"]), float(options["ray_end"])
    a.disparity_space_sampling = int(bool(options.get("disparity_space_sampling", False)))
    a.box_warp = float(options["box_warp"])
    a.white_back = int(bool(options.get("white_back", False)))
    if u_coarse is not None:
        u_coarse = _dev(u_coarse, "u_coarse").reshape(N, M, D)
        a.u_coarse = u_coarse.data_ptr()
    if u_fine is not None and Di > 0:
        u_fine = _dev(u_fine, "u_fine").reshape(N * M, Di)
        a.u_fine = u_fine.data_ptr()
    keep += [u_coarse, u_fine]
    if isinstance(seed, torch.Tensor):           # device-resident key (int64 [1]): graph-replay friendly
        assert seed.dtype == torch.int64 and seed.numel() == 1 and seed.device == dev
        a.seed_device = seed.data_ptr()
        keep.append(seed)
    else:
        a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    rgb = torch.empty((N, 32, M) if channels_first else (N, M, 32), device=dev)
    seg = torch.empty((N, 15, M) if channels_first else (N, M, 15), device=dev)
    depth = torch.empty(N, M, 1, device=dev)
    wsum = torch.empty(N, M, 1, device=dev)
    a.rgb, a.seg, a.depth, a.wsum = rgb.data_ptr(), seg.data_ptr(), depth.data_ptr(), wsum.data_ptr()
    a.channels_first = int(channels_first)
    tap = {}
    if taps:
        tap["depths_all"] = torch.empty(N, M, D + Di, device=dev)
        a.tap_depths_all = tap["depths_all"].data_ptr()
        if Di > 0:
            tap["weights_coarse"] = torch.empty(N, M, D - 1, device=dev)
            tap["depths_fine"] = torch.empty(N, M, Di, device=dev)
            a.tap_weights_coarse, a.tap_depths_fine = tap["weights_coarse"].data_ptr(), tap["depths_fine"].data_ptr()
        if sample_colors:
            # the decoders' outputs for every sample of the final march (192 bytes per sample, opaque): render_backward(...,
            # sample_colors=) then skips its re-evaluation pass.  Split-bf16 decoder without density noise / cross decoder only.
            assert a.decoder_math == _lib.NFE_MATH_BF16X3, "sample_colors needs the split-bf16 decoder"
            tap["sample_colors"] = torch.empty(lib.nfe_render_sample_colors_floats(N, M, D + Di), device=dev)
            a.tap_sample_colors = tap["sample_colors"].data_ptr()
            # the buffer's ray order follows the launch's ray-block shape (8x4 pixel tiles when resolution % 8 == 0 and resolution^2 ==
            # n_rays, else 32 consecutive rays): render_backward must be given the same `resolution` and checks it against this
            tap["sample_colors_resolution"] = int(a.resolution)
    need = lib.nfe_render_workspace_bytes(N, M, D, Di)
    if Di > 0 and a.density_noise > 0:                      # draw index of every merged sample (include/nfe_render.h)
        need += (N * M * (D + Di) * 4 + 255) // 256 * 256
    ws = _workspace(dev, need)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    with torch.cuda.device(dev):
        _lib.check(lib.nfe_render(ctypes.byref(a), _stream()), "nfe_render")
    return (rgb, seg, depth, wsum, tap) if taps else (rgb, seg, depth, wsum)


def render_backward(planes_geo, planes_app, decoder_heads, lr_mul, options, depths_all, grads, *, origins=None, dirs=None,
                    cam2world=None, intrinsics=None, resolution=0, affines=None, channels_first=False, need=(True, True),
                    sample_colors=None, sample_colors_resolution=None):
    """nfe_render_backward: the vector-Jacobian product of `render` w.r.t. the two plane sets (what autograd does for
    renderer.py:301-363 with the planes as leaves; depths are constants, renderer.py:198,211).

    planes_* packed [Np,3,H,W,32]; decoder_heads = the 8 raw decoder tensors (geo w0,b0,w1,b1, app w0,b0,w1,b1);
    depths_all [N,M,S] = the `depths_all` tap of the forward call; grads = (g_rgb, g_seg, g_depth, g_wsum), entries may
    be None.  Returns (grad_planes_geo, grad_planes_app) in gather layout [Np,3,H,W,32] (None where `need` is False; the
    same tensor twice when planes_app is planes_geo)."""
    lib = _lib.load()
    if float(options.get("density_noise", 0) or 0) > 0 and sample_colors is None:
        raise RuntimeError("render_backward: with density_noise > 0 the backward needs the forward's kept per-sample outputs (render(..., taps=True, "
                           "sample_colors=True) -> sample_colors=): it has no noise draws of its own to re-evaluate the samples with")
    planes_geo = _dev(planes_geo, "planes_geo", (None, 3, None, None, 32))
    same = planes_app is planes_geo or planes_app.data_ptr() == planes_geo.data_ptr()
    planes_app = planes_geo if same else _dev(planes_app, "planes_app", tuple(planes_geo.shape))
    Np, _, H, W, _ = planes_geo.shape
    dev = planes_geo.device
    depths_all = _dev(depths_all, "depths_all", (None, None, None))
    N, M, S = depths_all.shape
    assert Np in (1, N), f"planes batch {Np} must be 1 or equal the ray batch {N}"
    a = _lib.RenderBackwardArgs()
    a.struct_size = ctypes.sizeof(_lib.RenderBackwardArgs)
    a.planes_geo, a.planes_app = planes_geo.data_ptr(), planes_app.data_ptr()
    a.plane_h, a.plane_w = H, W
    bcast = Np == 1 and N > 1
    a.plane_view_stride = 0 if bcast else 3 * H * W * 32
    keep = [planes_geo, planes_app, depths_all]
    if affines is not None:
        affines = [_dev(t, "affine", (N, 96)) for t in affines]
        a.geo_scale, a.geo_shift, a.app_scale, a.app_shift = [t.data_ptr() for t in affines]
        keep += affines
    names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
             "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    heads = [_dev(t, nm, sh) for t, nm, sh in zip(decoder_heads, names, shapes)]
    (a.geo_w0, a.geo_b0, a.geo_w1, a.geo_b1, a.app_w0, a.app_b0, a.app_w1, a.app_b1) = [t.data_ptr() for t in heads]
    a.lr_mul = float(lr_mul)
    a.n_views, a.n_rays, a.n_samples = N, M, S
    if origins is not None:
        origins = _dev(origins, "ray_origins", (N, M, 3))
        dirs = _dev(dirs, "ray_directions", (N, M, 3))
        a.origins, a.dirs = origins.data_ptr(), dirs.data_ptr()
        r = int(round(M ** 0.5))                 # an r x r image in row-major order (RaySampler's order): lets the scatter kernel
        a.resolution = int(resolution) if resolution else (r if r * r == M else 0)     # group 8x8 neighbouring rays per wave
    else:
        cam2world = _dev(cam2world, "cam2world_matrix", (N, 4, 4))
        intrinsics = _dev(intrinsics, "intrinsics", (N, 3, 3))
        a.cam2world, a.intrinsics, a.resolution = cam2world.data_ptr(), intrinsics.data_ptr(), int(resolution)
    keep += [origins, dirs, cam2world, intrinsics] + heads
    a.depths = depths_all.data_ptr()
    a.box_warp = float(options["box_warp"])
    a.white_back = int(bool(options.get("white_back", False)))
    g_rgb, g_seg, g_depth, g_wsum = grads
    if g_rgb is not None:
        g_rgb = _dev(g_rgb, "grad_rgb", (N, 32, M) if channels_first else (N, M, 32)); a.grad_rgb = g_rgb.data_ptr()
    if g_seg is not None:
        g_seg = _dev(g_seg, "grad_seg", (N, 15, M) if channels_first else (N, M, 15)); a.grad_seg = g_seg.data_ptr()
    if g_depth is not None:
        g_depth = _dev(g_depth, "grad_depth").reshape(N, M); a.grad_depth = g_depth.data_ptr()
    if g_wsum is not None:
        g_wsum = _dev(g_wsum, "grad_wsum").reshape(N, M); a.grad_wsum = g_wsum.data_ptr()
    keep += [g_rgb, g_seg, g_depth, g_wsum]
    a.channels_first = int(channels_first)
    need_g, need_a = bool(need[0]), bool(need[1])
    gg = ga = None
    if same:
        gg = ga = torch.zeros_like(planes_geo) if (need_g or need_a) else None
    else:
        gg = torch.zeros_like(planes_geo) if need_g else None
        ga = torch.zeros_like(planes_app) if need_a else None
    if gg is None and ga is None:
        return None, None
    a.grad_planes_geo = gg.data_ptr() if gg is not None else None
    a.grad_planes_app = ga.data_ptr() if ga is not None else None
    a.grad_view_stride = 0 if bcast else 3 * H * W * 32
    if sample_colors is not None:           # the `sample_colors` tap of the forward call: no re-evaluation pass
        if sample_colors_resolution is None:
            raise ValueError("render_backward: sample_colors needs sample_colors_resolution (the `sample_colors_resolution` tap of the "
                             "forward call): the buffer's ray order depends on the forward launch's resolution")
        if int(sample_colors_resolution) != int(a.resolution):
            raise ValueError(f"render_backward: sample_colors were stored by a forward launch with resolution={int(sample_colors_resolution)}, "
                             f"this call resolves to resolution={int(a.resolution)}: colours would be paired with the wrong rays "
                             "(pass the same `resolution` to render and render_backward)")
        sample_colors = _dev(sample_colors, "sample_colors", (lib.nfe_render_sample_colors_floats(N, M, S),))
        a.sample_colors = sample_colors.data_ptr()
        keep.append(sample_colors)
    ws = _workspace(dev, lib.nfe_render_backward_workspace_bytes(N, M, S))
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    with torch.cuda.device(dev):
        _lib.check(lib.nfe_render_backward(ctypes.byref(a), _stream()), "nfe_render_backward")
    return gg, ga


def point_query(planes_geo, planes_app, decoder_packed, coords, box_warp, affines=None, decoder_math=None, density_noise=0.0, seed=0,
                decoder_cross=None):
    """nfe_point_query: coords [N,P,3] -> dict(rgb [N,P,32], sigma [N,P,1], seg [N,P,15]).  density_noise > 0 adds
    N(0,1) * density_noise to sigma (renderer.py:285-286; Philox normals keyed by `seed` and the point index)."""
    lib = _lib.load()
    planes_geo = _dev(planes_geo, "planes_geo", (None, 3, None, None, 32))
    planes_app = planes_geo if planes_app is planes_geo else _dev(planes_app, "planes_app", tuple(planes_geo.shape))
    coords = _dev(coords, "coordinates", (None, None, 3))
    N, P = coords.shape[0], coords.shape[1]
    Np, _, H, W, _ = planes_geo.shape
    assert Np in (1, N)
    dev = coords.device
    aff = [None] * 4
    if affines is not None:
        aff = [_dev(t, "affine", (N, 96)) for t in affines]
    decoder_packed = _dev(decoder_packed, "decoder_packed", (_lib.NFE_DECODER_PACKED_FLOATS,))
    rgb = torch.empty(N, P, 32, device=dev)
    sigma = torch.empty(N, P, 1, device=dev)
    seg = torch.empty(N, P, 15, device=dev)
    if P == 0:                                   # empty query: nothing to launch (zero-size tensors have no storage)
        return {"rgb": rgb, "sigma": sigma, "seg": seg}
    stride = 0 if (Np == 1 and N > 1) else 3 * H * W * 32
    with torch.cuda.device(dev):
        _lib.check(lib.nfe_point_query(_ptr(planes_geo), _ptr(planes_app), H, W, stride, *[_ptr(t) for t in aff],
                                       _ptr(decoder_packed), _math_mode(decoder_math), _ptr(coords), N, P, float(box_warp),
                                       _ptr(rgb), _ptr(sigma), _ptr(seg), float(density_noise), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                       _ptr(_dev(decoder_cross, "decoder_cross", (_lib.NFE_DECODER_CROSS_FLOATS,)) if decoder_cross is not None else None), _stream()),
                   "nfe_point_query")
    return {"rgb": rgb, "sigma": sigma, "seg": seg}
