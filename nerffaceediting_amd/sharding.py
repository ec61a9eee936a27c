"""View sharding for batch-of-views rendering.

Unit = one (latent/planes, camera) pair -> one frame; views are independent, so V views are cut into
contiguous blocks of ceil(V/G) per rank (keeps frame order after the gather) and the only exchange
is one all-gather of finished frames (RCCL over xGMI when the backend is 'nccl'; gloo in CPU tests).
The reference has no multi-GPU inference path (gen_videos.py:122-147 is a sequential batch-1 loop);
its only precedent is the gather-by-broadcast loop in metrics/metric_utils.py:126-136.
"""
import torch
import torch.distributed as dist


def shard_range(n_views, rank, world_size):
    """Contiguous [start, stop) block of views owned by `rank`; trailing ranks may be short or empty."""
    per = -(-int(n_views) // int(world_size))
    start = min(rank * per, n_views)
    return start, min(start + per, n_views)


def all_gather_frames(local_frames, n_views, group=None):
    """local_frames [v_local, ...] (this rank's block, in order) -> [n_views, ...] on every rank.

    Uses one all_gather_into_tensor on padded equal-size blocks (one collective per job or chunk).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local_frames.shape[0] == n_views
        return local_frames
    world = dist.get_world_size(group)
    per = -(-int(n_views) // world)
    pad = per - local_frames.shape[0]
    block = local_frames
    if pad:
        block = torch.cat([local_frames, local_frames.new_zeros((pad,) + tuple(local_frames.shape[1:]))], 0)
    out = block.new_empty((world * per,) + tuple(block.shape[1:]))
    dist.all_gather_into_tensor(out, block.contiguous(), group=group)
    return out[:n_views]


def all_gather_frames_async(local_frames, n_views, group=None):
    """Same exchange, not waited for: returns (work, frames).  The collective runs on the backend's own stream
    (RCCL: overlapped with whatever the caller launches next); call work.wait() before reading `frames`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local_frames.shape[0] == n_views
        return None, local_frames
    world = dist.get_world_size(group)
    per = -(-int(n_views) // world)
    pad = per - local_frames.shape[0]
    block = local_frames
    if pad:
        block = torch.cat([local_frames, local_frames.new_zeros((pad,) + tuple(local_frames.shape[1:]))], 0)
    out = block.new_empty((world * per,) + tuple(block.shape[1:]))
    work = dist.all_gather_into_tensor(out, block.contiguous(), group=group, async_op=True)
    return work, out[:n_views]
