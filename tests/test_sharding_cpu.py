"""CPU, world_size 2 over gloo: view sharding + frame all-gather (the N>1 path of bench.py / SURVEY §8e)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerffaceediting_amd import sharding


def test_shard_range_covers_all_views_in_order():
    for V in (1, 2, 7, 8, 512, 513):
        for G in (1, 2, 3, 8):
            spans = [sharding.shard_range(V, r, G) for r in range(G)]
            flat = [i for a, b in spans for i in range(a, b)]
            assert flat == list(range(V)), (V, G, spans)
            assert max(b - a for a, b in spans) == -(-V // G)


def _worker(rank, world, port, V, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = sharding.shard_range(V, rank, world)
        local = torch.stack([torch.full((3, 4, 4), float(i)) for i in range(a, b)]) if b > a else torch.zeros(0, 3, 4, 4)
        out = sharding.all_gather_frames(local, V)
        ok = out.shape == (V, 3, 4, 4) and all(float(out[i, 0, 0, 0]) == i for i in range(V))
        works = [sharding.all_gather_frames_async(local + k, V) for k in range(3)]          # bench.py keeps two in flight
        for k, (w, o) in enumerate(works):
            w.wait()
            ok = ok and o.shape == (V, 3, 4, 4) and all(float(o[i, 0, 0, 0]) == i + k for i in range(V))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V", [4, 5])
def test_all_gather_frames_world2(V):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + V
    procs = [ctx.Process(target=_worker, args=(r, 2, port, V, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_all_gather_single_process_is_identity():
    x = torch.arange(24.0).reshape(2, 3, 2, 2)
    assert sharding.all_gather_frames(x, 2) is x
    w, y = sharding.all_gather_frames_async(x, 2)
    assert w is None and y is x
