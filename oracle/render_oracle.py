"""CPU oracle for the NeRFFaceEditing volumetric-rendering inference path (numpy, fp32).

TEST INFRASTRUCTURE ONLY.  This file is a plain-numpy restatement of the reference's
algorithm for the render core.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``nerffaceediting_amd``) never does and fails loudly when its HIP library is missing.

Parity pinning: every function below was checked in the build container against the
reference itself (imported from /root/reference on CPU, torch 2.10.0) by
``oracle/gen_golden.py``; the resulting input/output vectors are committed under
``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks this file against them.
The reference has no tests or golden vectors of its own.

Each function cites the reference file:line it restates (paths relative to the
reference root).  Arithmetic is fp32 everywhere, in the reference's operation order
where the order is observable.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def _f32(x):
    return np.asarray(x, dtype=F32)


# --------------------------------------------------------------------------------------
# camera_utils.py  (input generators for tests / bench)
# --------------------------------------------------------------------------------------
def _normalize_vecs(v):
    # training/volumetric_rendering/math_utils.py:33-37
    return v / np.linalg.norm(v, axis=-1, keepdims=True).astype(F32)


def create_cam2world_matrix(forward, origin):
    """camera_utils.py:118-137 — y-up, no roll."""
    forward = _normalize_vecs(_f32(forward))
    up = np.broadcast_to(_f32([0, 1, 0]), forward.shape)
    right = -_normalize_vecs(np.cross(up, forward).astype(F32))
    up = _normalize_vecs(np.cross(forward, right).astype(F32))
    n = forward.shape[0]
    rot = np.tile(np.eye(4, dtype=F32), (n, 1, 1))
    rot[:, :3, :3] = np.stack((right, up, forward), axis=-1)
    trans = np.tile(np.eye(4, dtype=F32), (n, 1, 1))
    trans[:, :3, 3] = origin
    return (trans @ rot).astype(F32)


def lookat_pose(horizontal, vertical, lookat, radius=1.0):
    """camera_utils.py:69-86 (LookAtPoseSampler.sample with zero stddev) -> [1,4,4]."""
    h = F32(horizontal)
    v = F32(min(max(vertical, 1e-5), math.pi - 1e-5))
    theta = h
    v = F32(v / F32(math.pi))
    phi = np.arccos(F32(1) - F32(2) * v).astype(F32)
    o = np.zeros((1, 3), dtype=F32)
    o[:, 0] = F32(radius) * np.sin(phi) * np.cos(F32(math.pi) - theta)
    o[:, 2] = F32(radius) * np.sin(phi) * np.sin(F32(math.pi) - theta)
    o[:, 1] = F32(radius) * np.cos(phi)
    fwd = _normalize_vecs(_f32(lookat)[None, :] - o)
    return create_cam2world_matrix(fwd, o)


def fov_to_intrinsics(fov_degrees):
    """camera_utils.py:140-149 (note the reference's truncated pi and sqrt(2))."""
    focal = float(1 / (math.tan(fov_degrees * 3.14159 / 360) * 1.414))
    return _f32([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1]])


def make_c(cam2world, intrinsics):
    """25-d camera label = 16 pose | 9 intrinsics (gen_samples.py:171)."""
    cam2world = _f32(cam2world).reshape(-1, 16)
    intr = np.broadcast_to(_f32(intrinsics).reshape(-1, 9), (cam2world.shape[0], 9))
    return np.concatenate([cam2world, intr], axis=1).astype(F32)


# --------------------------------------------------------------------------------------
# Philox4x32-10: the build's production jitter source (the reference uses torch's global
# RNG, renderer.py:190,237; parity tests inject u instead).  Restated here so tests can
# check the in-kernel generator bit-for-bit.
# --------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PH_M0
            p1 = c2.astype(np.uint64) * _PH_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PH_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PH_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_uniform(n_rays, n_samples, seed, stream):
    """u[ray, k] in [0,1) with 24 random bits: counter=(ray, k//4, stream, 0), key=seed."""
    ray = np.arange(n_rays, dtype=np.uint32)[:, None]
    grp = np.arange((n_samples + 3) // 4, dtype=np.uint32)[None, :]
    out = philox4x32(ray, grp, np.uint32(stream), np.uint32(0),
                     seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    bits = np.stack(out, axis=-1).reshape(n_rays, -1)[:, :n_samples]
    return ((bits >> np.uint32(8)).astype(F32) * F32(1.0 / 16777216.0)).astype(F32)


# --------------------------------------------------------------------------------------
# a2  RaySampler
# --------------------------------------------------------------------------------------
def ray_sampler(cam2world, intrinsics, resolution):
    """training/volumetric_rendering/ray_sampler.py:24-62.

    cam2world [N,4,4], intrinsics [N,3,3] -> origins [N,R*R,3], dirs [N,R*R,3].
    Pixel m = i*R + j (row i, col j); x from the column, y from the row (:41-46).
    """
    cam2world = _f32(cam2world)
    intrinsics = _f32(intrinsics)
    N, R = cam2world.shape[0], int(resolution)
    fx, fy = intrinsics[:, 0, 0:1], intrinsics[:, 1, 1:2]
    cx, cy = intrinsics[:, 0, 2:3], intrinsics[:, 1, 2:3]
    sk = intrinsics[:, 0, 1:2]
    lin = np.arange(R, dtype=F32) * F32(1.0 / R) + F32(0.5 / R)           # :41
    ii, jj = np.meshgrid(lin, lin, indexing="ij")
    x_cam = np.broadcast_to(jj.reshape(1, -1), (N, R * R))                # flip(0): u <- col
    y_cam = np.broadcast_to(ii.reshape(1, -1), (N, R * R))
    x_lift = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx            # :50
    y_lift = (y_cam - cy) / fy                                            # :51
    ones = np.ones_like(x_lift)
    pts = np.stack((x_lift, y_lift, ones, ones), axis=-1).astype(F32)      # [N,M,4]
    world = np.einsum("nij,nmj->nmi", cam2world, pts).astype(F32)[..., :3]  # :55
    cam_loc = cam2world[:, :3, 3]
    d = world - cam_loc[:, None, :]
    nrm = np.sqrt((d * d).sum(-1, keepdims=True, dtype=F32))
    d = d / np.maximum(nrm, F32(1e-12))                                   # F.normalize :59
    o = np.broadcast_to(cam_loc[:, None, :], d.shape).copy()
    return o.astype(F32), d.astype(F32)


# --------------------------------------------------------------------------------------
# a4  plane statistics  (training/triplane.py:56-68)
# --------------------------------------------------------------------------------------
def compute_mean_var(planes):
    """triplane.py:56-60: mean and sqrt(unbiased var) over (H,W), keepdim."""
    planes = _f32(planes)
    mean = planes.mean(axis=(-1, -2), keepdims=True, dtype=np.float64)
    var = planes.astype(np.float64).var(axis=(-1, -2), keepdims=True, ddof=1)
    return mean.astype(F32), np.sqrt(var).astype(F32)


def normalize_plane(planes):
    """triplane.py:61-65."""
    mean, std = compute_mean_var(planes)
    return ((_f32(planes) - mean) / (std + F32(1e-8))).astype(F32), mean, std


def denormalize_plane(planes, mean, std):
    """triplane.py:66-68."""
    return (_f32(planes) * _f32(std) + _f32(mean)).astype(F32)


def synthesis_planes(planes, planes_mean=None, planes_var=None):
    """triplane.py:93-115: returns (norm_planes[N,3,32,H,W], denorm_planes[N,3,32,H,W], mean, std).

    planes_mean/var: None, tensors broadcastable to [N,96,1,1], or (int,int) batch indices
    into this batch's own statistics (:100-101).
    """
    planes = _f32(planes)
    norm, mean, std = normalize_plane(planes)
    if planes_mean is not None and planes_var is not None:
        if isinstance(planes_mean, int) and isinstance(planes_var, int):
            planes = denormalize_plane(norm, mean[planes_mean][None], std[planes_var][None])
        else:
            planes = denormalize_plane(norm, planes_mean, planes_var)
    N, _, H, W = planes.shape
    return norm.reshape(N, 3, 32, H, W), planes.reshape(N, 3, 32, H, W), mean, std


# --------------------------------------------------------------------------------------
# a5  tri-plane bilinear gather
# --------------------------------------------------------------------------------------
def project_onto_planes(coords):
    """renderer.py:23-53.  inv(plane_axes) are permutations: p0=(x,y), p1=(x,z), p2=(z,x)."""
    x, y, z = coords[..., 0], coords[..., 1], coords[..., 2]
    return np.stack([np.stack([x, y], -1), np.stack([x, z], -1), np.stack([z, x], -1)], axis=1)


def _grid_sample_bilinear_zeros(plane, gx, gy):
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False), renderer.py:64.

    plane [C,H,W]; gx indexes W, gy indexes H; returns [S,C].  Unnormalise as ATen's CPU
    kernel does: (g+1)*(size/2) - 0.5.
    """
    C, H, W = plane.shape
    ix = (gx + F32(1)) * F32(W / 2) - F32(0.5)
    iy = (gy + F32(1)) * F32(H / 2) - F32(0.5)
    x0f, y0f = np.floor(ix), np.floor(iy)
    dx, dy = (ix - x0f).astype(F32), (iy - y0f).astype(F32)
    ex, ey = F32(1) - dx, F32(1) - dy
    x0f = np.nan_to_num(x0f, nan=-10.0, posinf=1e9, neginf=-1e9)
    y0f = np.nan_to_num(y0f, nan=-10.0, posinf=1e9, neginf=-1e9)
    x0 = np.clip(x0f, -2, W + 1).astype(np.int64)
    y0 = np.clip(y0f, -2, H + 1).astype(np.int64)
    flat = np.ascontiguousarray(plane.reshape(C, H * W).T)               # [H*W, C]
    out = np.zeros((gx.shape[0], C), dtype=F32)
    for (xx, yy, ww) in ((x0, y0, ex * ey), (x0 + 1, y0, dx * ey),
                         (x0, y0 + 1, ex * dy), (x0 + 1, y0 + 1, dx * dy)):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        idx = np.where(ok, yy * W + xx, 0)
        out += flat[idx] * np.where(ok, ww, F32(0)).astype(F32)[:, None]
    return out


def sample_from_planes(planes, coords, box_warp):
    """renderer.py:55-65.  planes [N,3,C,H,W], coords [N,S,3] -> [N,3,S,C]."""
    planes = _f32(planes)
    coords = (F32(2.0 / box_warp) * _f32(coords)).astype(F32)             # :61
    proj = project_onto_planes(coords)                                   # [N,3,S,2]
    N, P = planes.shape[:2]
    out = np.empty((N, P, coords.shape[1], planes.shape[2]), dtype=F32)
    for n in range(N):
        for p in range(P):
            out[n, p] = _grid_sample_bilinear_zeros(planes[n, p], proj[n, p, :, 0], proj[n, p, :, 1])
    return out


# --------------------------------------------------------------------------------------
# a6  decoders
# --------------------------------------------------------------------------------------
def softplus(x, threshold=20.0):
    """torch.nn.Softplus(beta=1, threshold=20) (triplane.py:239,245) / F.softplus (ray_marcher.py:76)."""
    x = _f32(x)
    with np.errstate(over="ignore"):
        return np.where(x > F32(threshold), x, np.log1p(np.exp(np.minimum(x, F32(threshold))))).astype(F32)


def fully_connected(x, weight, bias, lr_mul=1.0):
    """networks_stylegan2.py:114-123 (activation='linear'): addmm(b*lr, x, (w*lr/sqrt(in)).T)."""
    w = (_f32(weight) * F32(lr_mul / np.sqrt(weight.shape[1]))).astype(F32)
    b = _f32(bias)
    if lr_mul != 1:
        b = b * F32(lr_mul)
    return (x @ w.T + b[None, :]).astype(F32)


def decoder_disentangled(f_norm, f_denorm, dec, lr_mul=1.0):
    """DisentangledOSGDecoder.forward, triplane.py:249-270.

    f_* [N,3,S,32]; dec = dict of geo_net.{0,2}.{weight,bias}, app_net.{0,2}.{weight,bias}.
    Returns rgb [N,S,32], sigma [N,S,1], seg [N,S,15].
    """
    fn = f_norm.mean(axis=1, dtype=F32)                                   # :251
    fd = f_denorm.mean(axis=1, dtype=F32)                                 # :252
    N, S, C = fn.shape
    h = softplus(fully_connected(fn.reshape(N * S, C), dec["geo_net.0.weight"], dec["geo_net.0.bias"], lr_mul))
    g = fully_connected(h, dec["geo_net.2.weight"], dec["geo_net.2.bias"], lr_mul).reshape(N, S, -1)
    sigma, seg = g[..., 0:1], g[..., 1:]                                  # :260-261
    h = softplus(fully_connected(fd.reshape(N * S, C), dec["app_net.0.weight"], dec["app_net.0.bias"], lr_mul))
    a = fully_connected(h, dec["app_net.2.weight"], dec["app_net.2.bias"], lr_mul).reshape(N, S, -1)
    with np.errstate(over="ignore"):
        rgb = (F32(1) / (F32(1) + np.exp(-a))) * F32(1 + 2 * 0.001) - F32(0.001)  # :269
    return rgb.astype(F32), sigma.astype(F32), seg.astype(F32)


def decoder_segmentation(f_denorm, dec, lr_mul=1.0):
    """SegmentationOSGDecoder.forward, triplane.py:209-230 (the `disable_alignment` ablation): sigma and rgb from `net`,
    seg from `seg_net`, BOTH on the denorm features.  dec = dict of net.{0,2}.{weight,bias}, seg_net.{0,2}.{weight,bias}."""
    fd = f_denorm.mean(axis=1, dtype=F32)                                 # :211
    N, S, C = fd.shape
    h = softplus(fully_connected(fd.reshape(N * S, C), dec["net.0.weight"], dec["net.0.bias"], lr_mul))
    x = fully_connected(h, dec["net.2.weight"], dec["net.2.bias"], lr_mul).reshape(N, S, -1)
    with np.errstate(over="ignore"):
        rgb = (F32(1) / (F32(1) + np.exp(-x[..., 1:]))) * F32(1 + 2 * 0.001) - F32(0.001)   # :219
    sigma = x[..., 0:1]
    h = softplus(fully_connected(fd.reshape(N * S, C), dec["seg_net.0.weight"], dec["seg_net.0.bias"], lr_mul))
    seg = fully_connected(h, dec["seg_net.2.weight"], dec["seg_net.2.bias"], lr_mul).reshape(N, S, -1)
    return rgb.astype(F32), sigma.astype(F32), seg.astype(F32)


def random_segmentation_decoder(seed, bias_scale=0.0):
    """Random SegmentationOSGDecoder parameters (state_dict names of the reference class)."""
    rng = np.random.RandomState(seed)
    dec = {}
    for net, outs in (("net", 33), ("seg_net", 15)):
        dec[f"{net}.0.weight"] = rng.randn(64, 32).astype(F32)
        dec[f"{net}.0.bias"] = (rng.randn(64) * bias_scale).astype(F32)
        dec[f"{net}.2.weight"] = rng.randn(outs, 64).astype(F32)
        dec[f"{net}.2.bias"] = (rng.randn(outs) * bias_scale).astype(F32)
    return dec


def run_model(norm_planes, denorm_planes, dec, coords, options, noise_seed=0):
    """DisentangledImportanceRenderer.run_model, renderer.py:259-287.  density_noise (:285-286; absent from every shipped
    config, train.py:288-323): the reference's randn_like stream is torch's; here the normal of point (n, m) is the Philox
    draw the HIP library makes (density_noise_normals with ray = n*P + m, draw 0)."""
    fn = sample_from_planes(norm_planes, coords, options["box_warp"])
    fd = sample_from_planes(denorm_planes, coords, options["box_warp"])
    if "seg_net.0.weight" in dec:
        rgb, sigma, seg = decoder_segmentation(fd, dec, options.get("decoder_lr_mul", 1))
    else:
        rgb, sigma, seg = decoder_disentangled(fn, fd, dec, options.get("decoder_lr_mul", 1))
    noise = F32(options.get("density_noise", 0) or 0)
    if noise > 0:
        N, P = coords.shape[:2]
        sigma = (sigma + noise * density_noise_normals(noise_seed, (N, P), np.zeros((N, P, 1), np.uint32))[..., 0]).astype(F32)
    return rgb, sigma, seg


# --------------------------------------------------------------------------------------
# a8  stratified depths
# --------------------------------------------------------------------------------------
def sample_stratified(N, M, ray_start, ray_end, D, u, disparity=False):
    """ImportanceRenderer.sample_stratified, renderer.py:169-192.  u [N,M,D] in [0,1).
    ray_start/ray_end: python scalars, or arrays [N,M,1] (the 'auto' branch, :183-186)."""
    u = _f32(u).reshape(N, M, D)
    if disparity:                                                         # :174-181
        lin = np.linspace(0, 1, D, dtype=F32).reshape(1, 1, D)
        s = lin + u * F32(1 / (D - 1))
        return (F32(1) / (F32(1.0 / ray_start) * (F32(1) - s) + F32(1.0 / ray_end) * s)).astype(F32)
    if isinstance(ray_start, np.ndarray):                                 # :183-186, math_utils.py:101-118
        rs, re = _f32(ray_start).reshape(N, M, 1), _f32(ray_end).reshape(N, M, 1)
        steps = (np.arange(D, dtype=F32) / F32(D - 1)).reshape(1, 1, D)
        t = rs + steps * (re - rs)
        return (t + u * ((re - rs) / F32(D - 1))).astype(F32)
    lin = np.linspace(ray_start, ray_end, D, dtype=F32).reshape(1, 1, D)   # :188
    delta = F32((ray_end - ray_start) / (D - 1))                          # :189
    return (lin + u * delta).astype(F32)                                  # :190


def get_ray_limits_box(origins, dirs, box_side_length):
    """math_utils.py:46-98 — slab test against the [-L/2, L/2]^3 box; (-1,-2) for misses."""
    o = _f32(origins).reshape(-1, 3)
    d = _f32(dirs).reshape(-1, 3)
    half = F32(box_side_length / 2)
    bounds = np.array([[-half] * 3, [half] * 3], dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F32(1) / d).astype(F32)
        sign = (inv < 0).astype(np.int64)
        valid = np.ones(o.shape[0], dtype=bool)
        tmin = (bounds[sign[:, 0], 0] - o[:, 0]) * inv[:, 0]
        tmax = (bounds[1 - sign[:, 0], 0] - o[:, 0]) * inv[:, 0]
        tymin = (bounds[sign[:, 1], 1] - o[:, 1]) * inv[:, 1]
        tymax = (bounds[1 - sign[:, 1], 1] - o[:, 1]) * inv[:, 1]
        valid[(tmin > tymax) | (tymin > tmax)] = False
        tmin = np.maximum(tmin, tymin)
        tmax = np.minimum(tmax, tymax)
        tzmin = (bounds[sign[:, 2], 2] - o[:, 2]) * inv[:, 2]
        tzmax = (bounds[1 - sign[:, 2], 2] - o[:, 2]) * inv[:, 2]
        valid[(tmin > tzmax) | (tzmin > tmax)] = False
        tmin = np.maximum(tmin, tzmin)
        tmax = np.minimum(tmax, tzmax)
    tmin = np.where(valid, tmin, F32(-1)).astype(F32)
    tmax = np.where(valid, tmax, F32(-2)).astype(F32)
    shp = origins.shape[:-1] + (1,)
    return tmin.reshape(shp), tmax.reshape(shp)


# --------------------------------------------------------------------------------------
# a9  ray marcher
# --------------------------------------------------------------------------------------
def ray_march(colors, segs, densities, depths, white_back=False, clamp=True):
    """SegMipRayMarcher2.run_forward, ray_marcher.py:68-101.

    colors [N,M,S,32], segs [N,M,S,15], densities [N,M,S,1], depths [N,M,S,1].
    The depth clamp uses min/max over the WHOLE depths tensor (:94); clamp=False skips it so a
    caller that chunks rays can apply the global bounds afterwards.
    Returns rgb [N,M,32], seg [N,M,15], depth [N,M,1], weights [N,M,S-1,1].
    """
    colors, segs, densities, depths = map(_f32, (colors, segs, densities, depths))
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    colors_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / F32(2)
    segs_mid = (segs[:, :, :-1] + segs[:, :, 1:]) / F32(2)
    dens_mid = (densities[:, :, :-1] + densities[:, :, 1:]) / F32(2)
    depths_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / F32(2)
    dens_mid = softplus(dens_mid - F32(1))                                # :76
    alpha = (F32(1) - np.exp(-(dens_mid * deltas))).astype(F32)           # :80-82
    shifted = np.concatenate([np.ones_like(alpha[:, :, :1]), F32(1) - alpha + F32(1e-10)], axis=-2)
    trans = np.cumprod(shifted, axis=-2, dtype=F32)[:, :, :-1]            # :85
    weights = (alpha * trans).astype(F32)
    rgb = (weights * colors_mid).sum(-2, dtype=F32)
    seg = (weights * segs_mid).sum(-2, dtype=F32)
    wtot = weights.sum(2, dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        depth = (weights * depths_mid).sum(-2, dtype=F32) / wtot
    depth = np.nan_to_num(depth, nan=np.inf, posinf=np.finfo(F32).max, neginf=np.finfo(F32).min)  # :93
    if clamp:
        depth = np.clip(depth, depths.min(), depths.max())                # :94
    depth = depth.astype(F32)
    if white_back:
        rgb = rgb + F32(1) - wtot                                         # :96-97
    rgb = rgb * F32(2) - F32(1)                                           # :99
    return rgb.astype(F32), seg.astype(F32), depth, weights


# --------------------------------------------------------------------------------------
# a10  importance sampling
# --------------------------------------------------------------------------------------
def sample_pdf(bins, weights, u, eps=1e-5):
    """ImportanceRenderer.sample_pdf, renderer.py:214-253.  bins [R,B+1], weights [R,B], u [R,Ni]."""
    bins, weights, u = _f32(bins), _f32(weights), _f32(u)
    R, B = weights.shape
    weights = weights + F32(eps)                                          # :228
    pdf = weights / weights.sum(-1, keepdims=True, dtype=F32)
    cdf = np.cumsum(pdf, axis=-1, dtype=F32)
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], axis=-1)       # [R,B+1]
    inds = np.empty(u.shape, dtype=np.int64)
    for r in range(R):                                                    # searchsorted(right=True) :240
        inds[r] = np.searchsorted(cdf[r], u[r], side="right")
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, B)
    cdf_b, cdf_a = np.take_along_axis(cdf, below, 1), np.take_along_axis(cdf, above, 1)
    bin_b, bin_a = np.take_along_axis(bins, below, 1), np.take_along_axis(bins, above, 1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < F32(eps), F32(1), denom)                     # :249
    return (bin_b + (u - cdf_b) / denom * (bin_a - bin_b)).astype(F32)    # :252


def sample_importance(z_vals, weights, n_importance, u):
    """ImportanceRenderer.sample_importance, renderer.py:194-212.

    z_vals [N,M,D,1], weights [N,M,D-1,1], u [N*M,Ni] -> [N,M,Ni,1].
    (The reference's .squeeze() at :206 breaks for N*M==1; not replicated.)
    """
    N, M, D, _ = z_vals.shape
    z = _f32(z_vals).reshape(N * M, D)
    w = _f32(weights).reshape(N * M, D - 1)
    neg = np.full((N * M, 1), -np.inf, dtype=F32)
    padded = np.concatenate([neg, w, neg], axis=1)                        # max_pool1d(k=2,s=1,pad=1) :205
    mp = np.maximum(padded[:, :-1], padded[:, 1:])                        # [R,D]
    w = ((mp[:, :-1] + mp[:, 1:]) * F32(0.5)).astype(F32)                 # avg_pool1d(k=2,s=1) :206 -> [R,D-1]
    w = w + F32(0.01)                                                     # :207
    z_mid = (F32(0.5) * (z[:, :-1] + z[:, 1:])).astype(F32)               # :209
    t = sample_pdf(z_mid, w[:, 1:-1], u)                                  # :210
    return t.reshape(N, M, n_importance, 1)


# --------------------------------------------------------------------------------------
# a11  merge, a12 orchestration
# --------------------------------------------------------------------------------------
def unify_samples(d1, c1, s1, den1, d2, c2, s2, den2):
    """DisentangledImportanceRenderer.unify_samples, renderer.py:288-300."""
    d = np.concatenate([d1, d2], axis=-2)
    c = np.concatenate([c1, c2], axis=-2)
    s = np.concatenate([s1, s2], axis=-2)
    den = np.concatenate([den1, den2], axis=-2)
    idx = np.argsort(d, axis=-2, kind="stable")
    take = lambda a: np.take_along_axis(a, np.broadcast_to(idx, a.shape[:-1] + (1,)) if a.shape[-1] == 1 else
                                        np.broadcast_to(idx, a.shape), axis=-2)
    return take(d), take(c), take(s), take(den)


def density_noise_normals(seed, n_rays_shape, draw):
    """N(0,1) per sample as the HIP kernel draws them: Philox counter (ray, draw index, stream 2), Box-Muller on the first
    two words.  draw [N,M,S] uint32: coarse sample k -> k, fine sample of ascending rank r -> D + r; ray = n*M + m."""
    N, M = n_rays_shape
    ray = np.arange(N * M, dtype=np.uint32).reshape(N, M, 1)
    draw = np.asarray(draw, dtype=np.uint32)
    x, y, _, _ = philox4x32(np.broadcast_to(ray, draw.shape), draw, np.uint32(2), np.uint32(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = ((x >> np.uint32(8)).astype(F32) + F32(0.5)) * F32(1.0 / 16777216.0)
    u2 = (y >> np.uint32(8)).astype(F32) * F32(1.0 / 16777216.0)
    z = np.sqrt(F32(-2.0) * np.log(u1)) * np.cos(F32(2.0 * np.pi) * u2)
    return z.astype(F32)[..., None]


def render(norm_planes, denorm_planes, dec, origins, dirs, options, u_coarse, u_fine=None,
           return_taps=False, clamp=True, noise_seed=0):
    """DisentangledImportanceRenderer.forward, renderer.py:301-363.

    norm_planes/denorm_planes [N,3,32,H,W]; origins/dirs [N,M,3]; u_coarse [N,M,D];
    u_fine [N*M,Ni] (needed when depth_resolution_importance>0).
    Returns rgb [N,M,32], seg [N,M,15], depth [N,M,1], wsum [N,M,1] (+ taps dict).
    """
    origins, dirs = _f32(origins), _f32(dirs)
    N, M, _ = origins.shape
    D = int(options["depth_resolution"])
    Ni = int(options.get("depth_resolution_importance", 0))
    wb = bool(options.get("white_back", False))
    assert options.get("clamp_mode", "softplus") == "softplus"
    if options["ray_start"] == "auto" and options["ray_end"] == "auto":    # :312-318
        rs, re = get_ray_limits_box(origins, dirs, options["box_warp"])
        ok = re > rs
        if ok.any():
            rs = np.where(ok, rs, rs[ok].min())
            re = np.where(ok, re, rs[ok].max())
        depths_c = sample_stratified(N, M, rs, re, D, u_coarse, options.get("disparity_space_sampling", False))
    else:
        depths_c = sample_stratified(N, M, options["ray_start"], options["ray_end"], D, u_coarse,
                                     options.get("disparity_space_sampling", False))
    depths_c = depths_c.reshape(N, M, D, 1)
    coords = (origins[:, :, None, :] + depths_c * dirs[:, :, None, :]).reshape(N, -1, 3)   # :326
    noise = F32(options.get("density_noise", 0) or 0)                   # renderer.py:285-286
    model_opts = {k: v for k, v in options.items() if k != "density_noise"}
    rgb_c, sig_c, seg_c = run_model(norm_planes, denorm_planes, dec, coords, model_opts)
    rgb_c = rgb_c.reshape(N, M, D, -1)
    sig_c = sig_c.reshape(N, M, D, 1)
    if noise > 0:
        sig_c = (sig_c + noise * density_noise_normals(noise_seed, (N, M), np.broadcast_to(np.arange(D, dtype=np.uint32), (N, M, D)))).astype(F32)
    seg_c = seg_c.reshape(N, M, D, -1)
    taps = {"depths_coarse": depths_c}
    if Ni > 0:
        _, _, _, w_c = ray_march(rgb_c, seg_c, sig_c, depths_c, wb)        # :340
        depths_f = sample_importance(depths_c, w_c, Ni, u_fine)           # :342
        coords = (origins[:, :, None, :] + depths_f * dirs[:, :, None, :]).reshape(N, -1, 3)
        rgb_f, sig_f, seg_f = run_model(norm_planes, denorm_planes, dec, coords, model_opts)
        sig_f = sig_f.reshape(N, M, Ni, 1)
        if noise > 0:
            rank = np.argsort(np.argsort(depths_f[..., 0], axis=-1, kind="stable"), axis=-1, kind="stable")   # ascending rank of each fine draw
            sig_f = (sig_f + noise * density_noise_normals(noise_seed, (N, M), (rank + D).astype(np.uint32))).astype(F32)
        all_d, all_c, all_s, all_den = unify_samples(
            depths_c, rgb_c, seg_c, sig_c, depths_f,
            rgb_f.reshape(N, M, Ni, -1), seg_f.reshape(N, M, Ni, -1), sig_f)
        rgb, seg, depth, w = ray_march(all_c, all_s, all_den, all_d, wb, clamp)   # :360
        taps.update(weights_coarse=w_c, depths_fine=depths_f, depths_all=all_d)
    else:
        rgb, seg, depth, w = ray_march(rgb_c, seg_c, sig_c, depths_c, wb, clamp)  # :362
    out = (rgb, seg, depth, w.sum(2, dtype=F32))
    return out + (taps,) if return_taps else out


def render_chunked(norm_planes, denorm_planes, dec, origins, dirs, options, u_coarse, u_fine=None,
                   chunk=8192):
    """render() over ray chunks; the whole-tensor depth clamp (ray_marcher.py:94) is applied once
    at the end with bounds taken over every chunk, so results equal the unchunked render()."""
    N, M, _ = origins.shape
    D = int(options["depth_resolution"])
    Ni = int(options.get("depth_resolution_importance", 0))
    outs, lo, hi = [], np.inf, -np.inf
    for m0 in range(0, M, chunk):
        m1 = min(M, m0 + chunk)
        uf = None
        if Ni > 0:
            uf = _f32(u_fine).reshape(N, M, Ni)[:, m0:m1].reshape(-1, Ni)
        r = render(norm_planes, denorm_planes, dec, origins[:, m0:m1], dirs[:, m0:m1], options,
                   _f32(u_coarse).reshape(N, M, D)[:, m0:m1], uf, return_taps=True, clamp=False)
        d_all = r[4].get("depths_all", r[4]["depths_coarse"])
        lo, hi = min(lo, float(d_all.min())), max(hi, float(d_all.max()))
        outs.append(r[:4])
    rgb, seg, depth, wsum = (np.concatenate([o[i] for o in outs], 1) for i in range(4))
    return rgb, seg, np.clip(depth, F32(lo), F32(hi)).astype(F32), wsum


# --------------------------------------------------------------------------------------
# a15  point query  (TriPlaneGenerator.sample / sample_mixed, triplane.py:140-157, minus the backbone)
# --------------------------------------------------------------------------------------
def point_query(planes, dec, coords, options):
    """planes [N,96,H,W] raw backbone output -> dict(rgb, sigma, seg) at coords [N,P,3]."""
    norm, denorm, _, _ = synthesis_planes(planes)
    rgb, sigma, seg = run_model(norm, denorm, dec, _f32(coords), options)
    return {"rgb": rgb, "sigma": sigma, "seg": seg}


# --------------------------------------------------------------------------------------
# helpers shared by tests / bench
# --------------------------------------------------------------------------------------
DEC_SHAPES = {
    "geo_net.0.weight": (64, 32), "geo_net.0.bias": (64,),
    "geo_net.2.weight": (16, 64), "geo_net.2.bias": (16,),
    "app_net.0.weight": (64, 32), "app_net.0.bias": (64,),
    "app_net.2.weight": (32, 64), "app_net.2.bias": (32,),
}


def random_decoder(seed, bias_scale=0.0):
    """Random-init decoder as FullyConnectedLayer does (randn weights / lr_mul, bias 0;
    networks_stylegan2.py:108-109); bias_scale>0 gives non-zero biases for stronger tests."""
    rng = np.random.RandomState(seed)
    dec = {}
    for k, shp in DEC_SHAPES.items():
        if k.endswith("weight"):
            dec[k] = rng.randn(*shp).astype(F32)
        else:
            dec[k] = (rng.randn(*shp) * bias_scale).astype(F32)
    return dec


FFHQ_OPTIONS = dict(  # train.py:288-313
    disparity_space_sampling=False, clamp_mode="softplus", depth_resolution=48,
    depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
    avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2], decoder_lr_mul=1,
)
