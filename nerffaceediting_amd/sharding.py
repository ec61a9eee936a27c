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


def _collective_active(group, force):
    """A world of one rank normally short-circuits every exchange; `force` keeps the collective (a one-rank RCCL group is all a
    1-GPU box can offer: it still exercises communicator init, the uint8 all_gather_into_tensor, the staging-buffer reuse and
    the stream ordering against the render streams - tests/test_rccl_single_gpu.py, bench.py --force-collective)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or bool(force)


def all_gather_frames(local_frames, n_views, group=None, force=False):
    """local_frames [v_local, ...] (this rank's block, in order) -> [n_views, ...] on every rank.

    Uses one all_gather_into_tensor on padded equal-size blocks (one collective per job or chunk).
    """
    if not _collective_active(group, force):
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


def all_gather_frames_async(local_frames, n_views, group=None, force=False):
    """Same exchange, not waited for: returns (work, frames).  The collective runs on the backend's own stream
    (RCCL: overlapped with whatever the caller launches next); call work.wait() before reading `frames`."""
    if not _collective_active(group, force):
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


class ChunkedFrameGather:
    """The exchange of a sharded job done chunk by chunk, overlapped with rendering ("one all_gather per
    chunk, overlapped with rendering").  Every rank owns the contiguous block shard_range(n_views, rank, world) and renders
    it `chunk` frames at a time; after round k it calls submit(k, frames_of_round_k) and goes on rendering round k+1 while
    the collective of round k runs on the backend's stream (RCCL over xGMI under 'nccl').  At most `max_in_flight`
    collectives are outstanding; a retired round is copied from its staging buffer [world, chunk, ...] to its final rows
    of the result [n_views, ...] (rank r's round k lands at start_r + k*chunk).  Ragged blocks (short or empty last
    rounds, empty ranks) are padded inside the staging buffer only.  finish() drains and returns the result.

    All ranks must call submit() for the same rounds in the same order: rounds() gives that count (from the longest
    block), and a rank whose block is exhausted submits an empty tensor.
    """

    def __init__(self, n_views, chunk, frame_shape, dtype, device, group=None, max_in_flight=2, force_collective=False):
        self.n_views, self.chunk, self.group = int(n_views), int(chunk), group
        self.active = _collective_active(group, force_collective)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.frame_shape, self.dtype, self.device = tuple(frame_shape), dtype, device
        self.spans = [shard_range(self.n_views, r, self.world) for r in range(self.world)]
        self.out = torch.empty((self.n_views,) + self.frame_shape, dtype=dtype, device=device)
        self.max_in_flight = max(1, int(max_in_flight))
        self.pending = []                      # (work, round, staging)
        self.free = []                         # staging buffers to reuse

    def rounds(self):
        longest = max(b - a for a, b in self.spans)
        return -(-longest // self.chunk)

    def local_slice(self, k):
        """[start, stop) of the frames this rank renders in round k (may be empty)."""
        a, b = self.spans[self.rank]
        s = min(a + k * self.chunk, b)
        return s, min(s + self.chunk, b)

    def _retire(self, entry):
        work, k, staging = entry
        if work is not None:
            work.wait()
        for r, (a, b) in enumerate(self.spans):
            s = min(a + k * self.chunk, b)
            e = min(s + self.chunk, b)
            if e > s:
                self.out[s:e].copy_(staging[r, :e - s])
        self.free.append(staging)

    def submit(self, k, frames):
        s, e = self.local_slice(k)
        assert frames.shape[0] == e - s and tuple(frames.shape[1:]) == self.frame_shape and frames.dtype == self.dtype, \
            (tuple(frames.shape), (e - s,) + self.frame_shape)
        if not self.active:
            if e > s:
                self.out[s:e].copy_(frames)
            return
        while len(self.pending) >= self.max_in_flight:
            self._retire(self.pending.pop(0))
        staging = self.free.pop() if self.free else torch.empty((self.world, self.chunk) + self.frame_shape, dtype=self.dtype, device=self.device)
        block = frames
        if e - s < self.chunk:                 # ragged tail / exhausted block: pad to the common chunk size
            block = frames.new_zeros((self.chunk,) + self.frame_shape)
            if e > s:
                block[:e - s].copy_(frames)
        work = dist.all_gather_into_tensor(staging.view((self.world * self.chunk,) + self.frame_shape), block.contiguous(),
                                           group=self.group, async_op=True)
        self.pending.append((work, k, staging))

    def finish(self):
        while self.pending:
            self._retire(self.pending.pop(0))
        return self.out
