"""Batch-of-views drivers: the counterparts of the reference's inference scripts for the path this package
accelerates.  gen_samples.py:160-183 renders 3 fixed yaws per seed, gen_videos.py:122-147 an
orbit of `w_frames` cameras — both sequentially, batch 1, on one GPU.  Here views are rendered in batches and
sharded over the ranks of a torch.distributed job (views are independent; one all-gather of frames).

File output (PNG / mp4 / mrc / ply) is out of scope; these return tensors.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import camera_utils, sharding

FFHQ_INTRINSICS = [[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]           # gen_videos.py:132


def seed_to_z(seed, z_dim=512, device="cuda"):
    """gen_samples.py:162 / gen_videos.py:84: z = RandomState(seed).randn(1, z_dim)."""
    return torch.from_numpy(np.random.RandomState(seed).randn(1, z_dim)).to(device=device, dtype=torch.float32)


def sample_cameras(device, cam_pivot=(0, 0, 0.2), cam_radius=2.7, fov_deg=18.837):
    """The three poses gen_samples.py:165-171 renders per seed -> c [3,25]."""
    intr = camera_utils.FOV_to_intrinsics(fov_deg, device=device)
    cs = []
    for yaw, pitch in [(0.4, -0.2), (0.0, -0.2), (-0.4, -0.2)]:
        c2w = camera_utils.LookAtPoseSampler.sample(np.pi / 2 + yaw, np.pi / 2 + pitch, torch.tensor(cam_pivot, device=device),
                                                    radius=cam_radius, device=device)
        cs.append(torch.cat([c2w.reshape(-1, 16), intr.reshape(-1, 9)], 1))
    return torch.cat(cs, 0)


def orbit_cameras(num_frames, device, lookat=(0, 0, 0.2), radius=2.7, pitch_range=0.25, yaw_range=0.35):
    """gen_videos.py:126-133 camera path (including its 3.14 constants) -> c [num_frames,25]."""
    intr = torch.tensor(FFHQ_INTRINSICS, device=device)
    cs = []
    for f in range(num_frames):
        c2w = camera_utils.LookAtPoseSampler.sample(3.14 / 2 + yaw_range * np.sin(2 * 3.14 * f / num_frames),
                                                    3.14 / 2 - 0.05 + pitch_range * np.cos(2 * 3.14 * f / num_frames),
                                                    torch.tensor(lookat, device=device), radius=radius, device=device)
        cs.append(torch.cat([c2w.reshape(-1, 16), intr.reshape(-1, 9)], 1))
    return torch.cat(cs, 0)


def interpolate_ws(ws_keyframes, w_frames=240, kind="cubic", wraps=2):
    """gen_videos.py:103-113,135-136: the per-frame latents of an interpolation video.  ws_keyframes [K,num_ws,w_dim] ->
    [K * w_frames, num_ws, w_dim]: frame f evaluates scipy's interp1d (cubic by default) of the keyframes tiled
    2*wraps+1 times (so the curve is periodic) at f / w_frames."""
    import scipy.interpolate
    K = ws_keyframes.shape[0]
    x = np.arange(-K * wraps, K * (wraps + 1))
    y = np.tile(ws_keyframes.detach().cpu().numpy(), [wraps * 2 + 1, 1, 1])
    interp = scipy.interpolate.interp1d(x, y, kind=kind, axis=0)
    out = interp(np.arange(K * w_frames) / w_frames)
    return torch.from_numpy(out).to(device=ws_keyframes.device, dtype=torch.float32)


@torch.no_grad()
def interpolation_video_frames(G, seeds, w_frames=240, kind="cubic", wraps=2, psi=1.0, truncation_cutoff=14, batch=4, gather=True,
                               image_mode="image", **synthesis_kwargs):
    """gen_videos.gen_interp_video (gen_videos.py:74-160) for a 1x1 grid, without the mp4 writer: keyframe latents from `seeds`
    (mapped under the frontal conditioning camera, :95-98), cubic interpolation in w, one orbit camera per frame (:126-133),
    frames rendered in batches and sharded over the ranks -> uint8 [F,512,512,3] (image_mode 'image' / 'image_raw')."""
    dev = next(G.parameters()).device
    zs = torch.cat([seed_to_z(s, G.z_dim, dev) for s in seeds], 0)
    c2w = camera_utils.LookAtPoseSampler.sample(3.14 / 2, 3.14 / 2, torch.tensor([0, 0, 0.2], device=dev), radius=2.7, device=dev)
    c_front = torch.cat([c2w.reshape(-1, 16), torch.tensor(FFHQ_INTRINSICS, device=dev).reshape(-1, 9)], 1).repeat(len(seeds), 1)
    ws_key = G.mapping(zs, c_front, truncation_psi=psi, truncation_cutoff=truncation_cutoff)
    ws = interpolate_ws(ws_key, w_frames, kind, wraps)
    c = orbit_cameras(ws.shape[0], dev)
    return render_views(G, ws, c, batch=batch, gather=gather, uint8=True, image_mode=image_mode, noise_mode="const", **synthesis_kwargs)


def to_uint8(img):
    """gen_samples.py:177: (img.permute(0,2,3,1) * 127.5 + 128).clamp(0,255).uint8."""
    return (img.permute(0, 2, 3, 1) * 127.5 + 128).clamp(0, 255).to(torch.uint8)


class StreamRing:
    """Issue consecutive batches on alternating HIP streams: the latency-bound volume render of one batch then overlaps the
    MFMA-bound convolutions of the next (measured +8 % views/s on the FFHQ configuration, tools/time_streams.py).
    run(fn) executes fn() on the next stream of the ring and returns (result, event); the consumer stream calls
    take(result, event) before touching the result (stream-ordered wait, no host sync)."""

    def __init__(self, device, n_streams=3):
        self.main = torch.cuda.current_stream(device)
        self.streams = [torch.cuda.Stream(device) for _ in range(n_streams)] if n_streams > 1 else []
        for s in self.streams:
            s.wait_stream(self.main)          # everything already queued (weights, inputs) is visible to the ring
        self.i = 0

    def run(self, fn):
        if not self.streams:
            return fn(), None
        s = self.streams[self.i % len(self.streams)]
        self.i += 1
        with torch.cuda.stream(s):
            out = fn()
            ev = torch.cuda.Event()
            ev.record(s)
        return out, ev

    def take(self, out, ev):
        if ev is not None:
            self.main.wait_event(ev)
            for t in (out.values() if isinstance(out, dict) else [out]):
                if isinstance(t, torch.Tensor):
                    t.record_stream(self.main)
        return out

    def drain(self):
        for s in self.streams:
            self.main.wait_stream(s)


@torch.no_grad()
def render_views(G, ws, c, batch=4, gather=True, uint8=False, image_mode="image", overlap=True, streams=3, **synthesis_kwargs):
    """Render V independent (ws[v], c[v]) pairs -> frames in view order on every rank: fp32 [V,3,H,W], or with
    uint8=True the gen_samples.py:177 conversion [V,H,W,3] (4x fewer bytes on the wire).

    ws [V,num_ws,w_dim] (or [1,...] broadcast: one identity, many cameras, as utils.render_video does), c [V,25].
    Under torch.distributed each rank renders its contiguous block of views (sharding.shard_range) `batch` frames at a
    time and the frames are exchanged chunk by chunk (sharding.ChunkedFrameGather): the all-gather of chunk k runs on the
    backend's stream under the rendering of chunk k+1 (overlap=False: one all-gather at the end).  gather=False returns
    this rank's block only."""
    V = c.shape[0]
    if ws.shape[0] == 1 and V > 1:
        ws = ws.expand(V, -1, -1)
    dev = c.device
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)

    # frame geometry per output key (triplane.py:131-138): `image` is the SR output, the others are neural-render sized
    channels = {"image": 3, "image_raw": 3, "image_seg": 15, "image_depth": 1}[image_mode]
    res = G.img_resolution if image_mode == "image" else (synthesis_kwargs.get("neural_rendering_resolution") or G.neural_rendering_resolution)
    frame_shape = (res, res, channels) if uint8 else (channels, res, res)

    def frames_of(i, j):
        if j <= i:
            return torch.zeros((0,) + frame_shape, dtype=torch.uint8 if uint8 else torch.float32, device=dev)
        img = G.synthesis(ws[i:j].contiguous(), c[i:j].contiguous(), **synthesis_kwargs)[image_mode]
        return to_uint8(img) if uint8 else img
    ring = StreamRing(dev, streams)
    if gather and world > 1 and overlap:
        gat = sharding.ChunkedFrameGather(V, batch, frame_shape, torch.uint8 if uint8 else torch.float32, dev)
        for k in range(gat.rounds()):
            sl = gat.local_slice(k)
            gat.submit(k, ring.take(*ring.run(lambda: frames_of(*sl))))
        return gat.finish()
    a, b = sharding.shard_range(V, rank, world)
    frames = [ring.take(*ring.run(lambda i=i: frames_of(i, min(b, i + batch)))) for i in range(a, b, batch)]
    local = torch.cat(frames, 0) if frames else frames_of(0, 0)
    return sharding.all_gather_frames(local, V) if (gather and world > 1) else local


def create_samples(N=256, voxel_origin=(0, 0, 0), cube_length=2.0, device=None):
    """gen_samples.py:79-101, bit for bit: the N^3 query points of the shape sweep, [1, N^3, 3], z fastest.  The reference
    divides the flat index as FLOATS (`overall_index.float() / N`), so only the z column sits on the lattice; the y and x
    columns advance by 1/N and 1/N^2 of a voxel per point.  Reproduced as is, in fp32, since the extracted volume is defined
    by these coordinates.  Returns (samples, voxel_origin, voxel_size) like the reference."""
    import numpy as np
    voxel_origin = np.array(voxel_origin) - cube_length / 2
    voxel_size = cube_length / (N - 1)
    idx = torch.arange(0, N ** 3, 1, dtype=torch.int64, device=device)
    f = idx.float()
    samples = torch.zeros(N ** 3, 3, device=device)
    samples[:, 2] = idx % N
    samples[:, 1] = (f / N) % N
    samples[:, 0] = ((f / N) / N) % N
    samples[:, 0] = (samples[:, 0] * voxel_size) + voxel_origin[2]
    samples[:, 1] = (samples[:, 1] * voxel_size) + voxel_origin[1]
    samples[:, 2] = (samples[:, 2] * voxel_size) + voxel_origin[0]
    return samples.unsqueeze(0), voxel_origin, voxel_size


@torch.no_grad()
def extract_density(G, ws, shape_res=128, max_batch=1 << 20, cube_length=None, **synthesis_kwargs):
    """gen_samples.py:186-203 shape extraction: sigma at the create_samples points -> [R,R,R] indexed [x,y,z] like the
    reference's `sigmas.reshape((shape_res,)*3)`.  The tri-planes are synthesised ONCE and reused for every chunk (the
    reference re-runs mapping + backbone per 10^6-point chunk)."""
    from . import ops
    from .training.triplane import packed_cross_of
    L = cube_length if cube_length is not None else G.rendering_kwargs["box_warp"] * 1
    R = shape_res
    pts, _, _ = create_samples(N=R, voxel_origin=[0, 0, 0], cube_length=L, device=ws.device)
    packed, mean, var = G._planes(ws, synthesis_kwargs)
    aff = ops.make_affine(mean, var)
    if G.disable_disentangle:                  # triplane.py:144-148: both heads read the raw planes
        aff = tuple(torch.ones_like(a) if i % 2 == 0 else torch.zeros_like(a) for i, a in enumerate(aff))
    out = torch.empty(R ** 3, device=ws.device)
    for s in range(0, R ** 3, max_batch):
        e = min(R ** 3, s + max_batch)
        out[s:e] = ops.point_query(packed, packed, G.decoder.packed(), pts[:, s:e].contiguous(), G.rendering_kwargs["box_warp"],
                                   affines=aff, decoder_math=G.renderer.decoder_math,
                                   decoder_cross=packed_cross_of(G.decoder))["sigma"].reshape(-1)
    return out.reshape(R, R, R)


def density_to_volume(sigmas, pad_value=-1000.0):
    """gen_samples.py:204-216: what is written to the .mrc / handed to marching cubes — flip axis 0 and overwrite a border of
    int(30 * shape_res / 256) voxels on every face with -1000."""
    R = sigmas.shape[0]
    vol = torch.flip(sigmas, dims=(0,)).clone()
    pad = int(30 * R / 256)
    if pad > 0:
        vol[:pad] = pad_value; vol[-pad:] = pad_value
        vol[:, :pad] = pad_value; vol[:, -pad:] = pad_value
        vol[:, :, :pad] = pad_value; vol[:, :, -pad:] = pad_value
    return vol
