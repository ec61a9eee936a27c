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


def _chunk_worker(rank, world, port, V, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.ChunkedFrameGather(V, chunk, (2, 3), torch.uint8, torch.device("cpu"), max_in_flight=2)
        order = []
        for k in range(g.rounds()):
            s, e = g.local_slice(k)
            order.append((s, e))
            frames = torch.stack([torch.full((2, 3), i, dtype=torch.uint8) for i in range(s, e)]) if e > s else torch.zeros(0, 2, 3, dtype=torch.uint8)
            g.submit(k, frames)                       # "render" of round k+1 proceeds while round k is exchanged
        out = g.finish()
        ok = out.shape == (V, 2, 3) and all(int(out[i, 1, 2]) == i for i in range(V))
        a, b = sharding.shard_range(V, rank, world)
        ok = ok and [i for s, e in order for i in range(s, e)] == list(range(a, b))      # every owned frame rendered once, in order
        q.put((rank, bool(ok), g.rounds()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V,chunk,rounds", [(5, 2, 2), (7, 3, 2), (3, 4, 1), (9, 2, 3)])
def test_chunked_overlapped_gather_world2_uneven_shards(V, chunk, rounds):
    """The chunked schedule of the sharded orbit (bench.py --workload orbit, apps.render_views): uneven blocks (V odd),
    ragged last rounds, a rank that runs out of frames before the other, more rounds than collectives in flight."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + V * 7 + chunk
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, V, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, rounds), (1, True, rounds)]


def test_chunked_gather_single_process():
    g = sharding.ChunkedFrameGather(5, 2, (1,), torch.float32, torch.device("cpu"))
    assert g.rounds() == 3 and not g.active
    for k in range(g.rounds()):
        s, e = g.local_slice(k)
        g.submit(k, torch.arange(s, e, dtype=torch.float32).reshape(-1, 1))
    assert g.finish().reshape(-1).tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_all_gather_single_process_is_identity():
    x = torch.arange(24.0).reshape(2, 3, 2, 2)
    assert sharding.all_gather_frames(x, 2) is x
    w, y = sharding.all_gather_frames_async(x, 2)
    assert w is None and y is x
