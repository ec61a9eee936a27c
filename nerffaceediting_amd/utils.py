"""demo.ipynb helpers with the reference's names (utils.py:32-88,146-199): encode / decode tri-planes,
normalise / denormalise them (appearance transfer = swapping mean/std between identities), and the orbit of
`render_video`.  mp4 encoding is out of scope: `render_video_frames` returns the uint8 frames and `render_video` (the reference's
signature) hands them to imageio when it is installed, or to a caller-supplied writer."""
import numpy as np
import torch
import torch.distributed as dist

from . import dense_ops, ops, sharding, sr_grad
from .camera_utils import FOV_to_intrinsics, LookAtPoseSampler


def _stats5(planes):
    N, P, C, H, W = planes.shape
    mean, var = ops.plane_stats(planes.reshape(N, P * C, H, W))
    return mean.reshape(N, P, C, 1, 1), var.reshape(N, P, C, 1, 1)


def compute_mean_var(planes):
    """utils.py:146-150 on (N,3,C,H,W) planes."""
    return _stats5(planes)


def normalize_plane(planes):
    """utils.py:152-155."""
    if _in_graph(planes):                                    # differentiable form, statistics included, as in the reference
        mean = planes.mean(dim=(-1, -2), keepdim=True)
        var = torch.sqrt(planes.var(dim=(-1, -2), keepdim=True))
        return (planes - mean) / (var + 1e-8), mean, var
    N, P, C, H, W = planes.shape
    mean, var = _stats5(planes)
    gs, gb, _, _ = ops.make_affine(mean.reshape(N, P * C, 1, 1), var.reshape(N, P * C, 1, 1))
    out = ops.plane_affine(planes.reshape(N, P * C, H, W), gs.reshape(N, P * C, 1, 1), gb.reshape(N, P * C, 1, 1))
    return out.reshape(N, P, C, H, W), mean, var


def _in_graph(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def denormalize_plane(planes, mean, var):
    """utils.py:157-158.  With a leaf among the arguments (plane editing) the affine is a torch expression, so gradients reach it."""
    if _in_graph(planes, mean, var):
        return planes * var + mean
    N, P, C, H, W = planes.shape
    out = ops.plane_affine(planes.reshape(N, P * C, H, W), var.reshape(-1, P * C, 1, 1).contiguous(),
                           mean.reshape(-1, P * C, 1, 1).contiguous())
    return out.reshape(N, P, C, H, W)


def _make_grid(img, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a [N,C,H,W] batch with N > 1 (the only form utils.render_tensor reaches it in): images left
    to right, `nrow` per row, `padding` pixels of `pad_value` around every image."""
    n, c, h, w = img.shape
    xmaps = min(int(nrow), n)
    ymaps = -(-n // xmaps)
    grid = img.new_full((c, (h + padding) * ymaps + padding, (w + padding) * xmaps + padding), pad_value)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[:, y * (h + padding) + padding:y * (h + padding) + padding + h, x * (w + padding) + padding:x * (w + padding) + padding + w] = img[k]
    return grid


@torch.no_grad()
def render_tensor(img, normalize=True, nrow=8):
    """utils.py:11-30: tensor (or list of [1,C,H,W] tensors) in [-1,1] -> PIL image; a batch becomes a grid of `nrow` columns
    (torchvision's make_grid layout, restated in _make_grid: torchvision is not a dependency of this package).  Same quirks as the
    reference: one-channel inputs are broadcast to three, the batch axis of a single image is squeezed away, values are scaled by 255
    and truncated (not rounded, not clamped) by the uint8 cast."""
    from PIL import Image
    if type(img) == list:
        img = torch.cat(img, dim=0).expand(-1, 3, -1, -1)
    elif len(img.shape) == 3:
        img = img.expand(3, -1, -1)
    elif len(img.shape) == 4:
        img = img.expand(-1, 3, -1, -1)
    img = img.squeeze()
    if normalize:
        img = img / 2 + .5
    if len(img.shape) == 3:
        return Image.fromarray((img.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8))
    if len(img.shape) == 2:
        return Image.fromarray((img.cpu().numpy() * 255).astype(np.uint8))
    return Image.fromarray((_make_grid(img, nrow=nrow).permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8))


class _NotDifferentiableImage(torch.autograd.Function):
    """Fallback for a super-resolution head class sr_grad.py does not know (sr_grad.supported() accepts every two-block head of the
    reference at any neural rendering resolution; anything else lands here): `image` is tied to the graph through this node so that
    a loss term on it fails loudly in backward() instead of silently contributing a zero plane gradient."""

    @staticmethod
    def forward(ctx, image, anchor):
        return image.view_as(image)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError("decode(): out['image'] is differentiable with respect to the planes only for the reference's own "
                           "super-resolution head classes (sr_grad.HEADS); for this head build the editing loss on image_raw / "
                           "image_seg / image_depth, or detach out['image'] explicitly")


def encode(G, ws, **synthesis_kwargs):
    """utils.py:160-163: ws -> planes (N,3,32,256,256)."""
    planes = G.backbone.synthesis(ws, **synthesis_kwargs)
    return planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])


def decode(G, ws, cam, norm_planes, denorm_planes, differentiable_image=True, **synthesis_kwargs):
    """utils.py:165-199: render (possibly edited) planes from camera(s) `cam` [N,25].  One plane set may serve
    several cameras (planes batch 1, N cameras).

    When a plane tensor requires grad, out['image'] is differentiable as in the reference; that costs memory: the SR head then runs
    layer by layer and keeps its six activations (about 0.4 GB per 512^2 view) until backward.  An editing loop whose loss never reads
    out['image'] (e.g. a segmentation loss on image_seg, as optimize_planes) can pass `differentiable_image=False`: the head runs
    fused, nothing is kept, and out['image'] is a detached tensor (an extension of the reference's signature; the default keeps the
    reference's behaviour).  The renderer's own kept per-sample colours (192 B per sample) are `G.renderer.keep_sample_colors`.
    On the differentiable path only `noise_mode` of the synthesis kwargs reaches the head (the reference's other block kwargs -
    `force_fp32`, `fused_modconv`, `update_emas` - select code paths this package does not have)."""
    cam2world_matrix = cam[:, :16].reshape(-1, 4, 4).contiguous()
    intrinsics = cam[:, 16:25].reshape(-1, 3, 3).contiguous()
    R = G.neural_rendering_resolution
    ray_origins, ray_directions = G.ray_sampler(cam2world_matrix, intrinsics, R)
    N = ray_origins.shape[0]
    feature_samples, seg_samples, depth_samples, _ = G.renderer(norm_planes, denorm_planes, G.decoder, ray_origins,
                                                                ray_directions, G.rendering_kwargs)
    feat = feature_samples.view(N, R, R, 32)                      # NHWC already
    rgb = feat[..., :3].contiguous()
    if ws.shape[0] == 1 and N > 1:
        ws = ws.expand(N, -1, -1).contiguous()
    sr_noise = G.rendering_kwargs["superresolution_noise_mode"]
    in_graph = torch.is_grad_enabled() and feature_samples.requires_grad
    if in_graph and differentiable_image and sr_grad.supported(G.superresolution, R):
        # planes are being optimised: the head runs layer by layer and keeps its activations, `image` carries the gradient
        # back to the feature image through the same MFMA kernels (sr_grad.py), as the reference's decode() does by autograd
        image = sr_grad.SRImage.apply(feat, G.superresolution, ws, sr_noise).permute(0, 3, 1, 2)
    else:
        sr = G.superresolution.forward_nhwc(rgb, feat, ws, noise_mode=sr_noise,
                                            **{k: v for k, v in synthesis_kwargs.items() if k != "noise_mode"})
        image = dense_ops.nhwc_to_nchw(sr)
        if in_graph and differentiable_image:    # an SR head class sr_grad does not know: no backward built
            image = _NotDifferentiableImage.apply(image, feature_samples)
    return {"image_raw": dense_ops.nhwc_to_nchw(rgb), "image": image,
            "image_depth": depth_samples.permute(0, 2, 1).reshape(N, 1, R, R),
            "image_seg": dense_ops.nhwc_to_nchw(seg_samples.view(N, R, R, 15))}


def video_camera_schedule(frames=150, a_degree=15.0, b_degree=12.0, init_pitch=5 * np.pi / 12, init_yaw=np.pi / 2):
    """The (pitch, yaw) list of utils.render_video (utils.py:45-73)."""
    frames_interp = frames // 4
    a, b = a_degree / 180 * np.pi, b_degree / 180 * np.pi
    start_pitch, start_yaw = np.pi / 2 - a, np.pi / 2
    sched = []
    if start_pitch != init_pitch:
        for index in range(frames_interp):
            ratio = index / (frames_interp - 1)
            sched.append((start_pitch * ratio + init_pitch * (1 - ratio), start_yaw * ratio + init_yaw * (1 - ratio)))
    for index in range(frames):
        theta = index / (frames - 1) * 2 * np.pi
        sched.append((np.pi / 2 - a * np.cos(theta), np.pi / 2 + b * np.sin(theta)))
    return sched


@torch.no_grad()
def render_video_frames(G, ws, norm_planes, denorm_planes, frames=150, a_degree=15.0, b_degree=12.0,
                        init_pitch=5 * np.pi / 12, init_yaw=np.pi / 2, batch=4):
    """utils.render_video (utils.py:32-88) without the mp4 writer: uint8 frames [F,512,512,3].  Cameras are
    batched, and sharded over torch.distributed ranks when a process group is up (one plane set, many cameras)."""
    dev = ws.device
    intrinsics = FOV_to_intrinsics(18.837, device=dev)
    pivot = torch.tensor(G.rendering_kwargs.get("avg_camera_pivot", [0, 0, 0]), device=dev, dtype=torch.float32)
    radius = G.rendering_kwargs.get("avg_camera_radius", 2.7)
    sched = video_camera_schedule(frames, a_degree, b_degree, init_pitch, init_yaw)
    cams = torch.cat([torch.cat([LookAtPoseSampler.sample(p, y, pivot, radius=radius, device=dev).reshape(-1, 16),
                                 intrinsics.reshape(-1, 9)], 1) for p, y in sched], 0)
    V = cams.shape[0]
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    a, b = sharding.shard_range(V, rank, world)
    out = []
    for i in range(a, b, batch):
        img = decode(G, ws, cams[i:min(b, i + batch)], norm_planes, denorm_planes, noise_mode="const")["image"]
        img = torch.round((img + 1) * (255 / 2)).clamp(0, 255).to(torch.uint8)          # utils.py:81-83
        out.append(img.permute(0, 2, 3, 1))
    local = torch.cat(out, 0) if out else torch.zeros(0, G.img_resolution, G.img_resolution, 3, dtype=torch.uint8, device=dev)
    return sharding.all_gather_frames(local, V) if world > 1 else local


@torch.no_grad()
def render_video(G, fn, ws, norm_planes, denorm_planes, frames=150, fps=30, a_degree=15.0, b_degree=12.0,
                 init_pitch=5 * np.pi / 12, init_yaw=np.pi / 2, writer=None, batch=4):
    """utils.render_video (utils.py:32-88), same positional signature.  The frames come from `render_video_frames` (cameras
    batched and sharded); rank 0 hands them, one HxWx3 uint8 array at a time, to `writer` — an object with
    `append_data(frame)` / `close()` like imageio's, or a callable `writer(frame)`.  With writer=None the reference's own
    `imageio.get_writer(fn, fps=fps, quality=8)` is used when imageio is importable; a `.npy` file name is written with numpy;
    anything else raises (mp4 encoding itself is not part of this package).  Returns the uint8 frames [F,H,W,3]."""
    import os
    out = render_video_frames(G, ws, norm_planes, denorm_planes, frames=frames, a_degree=a_degree, b_degree=b_degree,
                              init_pitch=init_pitch, init_yaw=init_yaw, batch=batch)
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return out
    if fn and os.path.dirname(fn):
        os.makedirs(os.path.dirname(fn), exist_ok=True)                # utils.py:75
    host = out.cpu().numpy()
    from . import ops
    ops.raise_if_handoff_lost()          # the copy above synchronised: a frame poisoned by a lost wave hand-off is an error here, not a black frame in the video
    if writer is None and fn is not None and str(fn).endswith(".npy"):
        np.save(fn, host)
        return out
    own = writer is None
    if own:
        try:
            import imageio
        except ImportError as e:
            raise RuntimeError("render_video(): imageio is not installed; pass writer=<object with append_data/close, or a "
                               "callable taking one HxWx3 uint8 frame>, or a '.npy' file name") from e
        writer = imageio.get_writer(fn, fps=fps, quality=8)           # utils.py:76
    put = writer.append_data if hasattr(writer, "append_data") else writer
    for frame in host:
        put(frame)
    if own:
        writer.close()
    return out


def get_camera_samples(G, device):
    """utils.py:130-144: the 3 x 3 grid of (pitch, yaw) in {5,6,7}*pi/12 demo.ipynb renders -> list of nine c [1,25]."""
    intrinsics = FOV_to_intrinsics(18.837, device=device)
    cam_pivot = torch.tensor(G.rendering_kwargs.get("avg_camera_pivot", [0, 0, 0]), device=device)
    cam_radius = G.rendering_kwargs.get("avg_camera_radius", 2.7)
    angles = [5 * np.pi / 12, 6 * np.pi / 12, 7 * np.pi / 12]
    return [torch.cat([LookAtPoseSampler.sample(pitch, yaw, cam_pivot, radius=cam_radius, device=device).reshape(-1, 16),
                       intrinsics.reshape(-1, 9)], 1) for pitch in angles for yaw in angles]


PART_COLORS = [[0, 0, 0], [127, 212, 255], [255, 212, 255], [255, 255, 170], [255, 255, 130], [76, 153, 0], [0, 255, 170],
               [244, 124, 244], [30, 162, 230], [127, 255, 255], [127, 170, 255], [85, 0, 255], [255, 170, 127], [212, 127, 255],
               [0, 170, 255], [255, 255, 255]]


@torch.no_grad()
def vis_parsing_maps(im, inverse=False, argmax=True):
    """utils.py:91-128: seg logits [N,15+,H,W] (or label map) -> colour image in [-1,1]; inverse: colours -> labels."""
    colors = torch.tensor(PART_COLORS, device=im.device, dtype=torch.float32)
    if not inverse:
        if argmax:
            im = torch.argmax(im, dim=1, keepdim=True)
        idx = im[:, 0].long().clamp(0, len(PART_COLORS) - 1)
        out = colors[idx].permute(0, 3, 1, 2)
        out = torch.where((im >= 0) & (im < len(PART_COLORS)), out, torch.zeros_like(out))
        return out / 255.0 * 2 - 1
    out = torch.zeros((im.size(0), 1, im.size(2), im.size(3)), device=im.device, dtype=torch.int64)
    for index in range(len(PART_COLORS)):
        color = colors[index].to(im.dtype).view(1, 3, 1, 1) / 255.0 * 2 - 1
        out = torch.where(torch.all((im - color).abs() <= 1e-2, dim=1, keepdim=True), torch.full_like(out, index), out)
    return out
